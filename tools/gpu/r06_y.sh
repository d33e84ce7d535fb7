#!/bin/bash
# round 6: the frame's constants in vector registers (-DSF_VREG_CONSTANTS) once more, on the x-row kernel: is a v_fma with a scalar source the 4.3-cycle instruction the issue table says?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06y
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-depth-only --no-single-frame --no-pmc --repeats 3"
for flag in "" "-DSF_VREG_CONSTANTS" "-DSF_VREG_CONSTANTS -DSF_INT_WAVES=4" "-DSF_INT_WAVES=4"; do
  touch scannet_amd/csrc/fuser.hip
  SCANFUSE_BUILD_FLAGS="$flag" python -c "from scannet_amd import build as b; b.build()" > $O/build.log 2>&1 || tail -5 $O/build.log
  python tools/kernel_resources.py 2>/dev/null | grep -E "k_integrate<1, 2, true, 2, false, 4, true" | cut -c1-150
  ( SF_BENCH_DETAIL=$O/detail.json timeout 300 python bench.py $Q ) > $O/bench.json 2> $O/bench.err
  python -c "
import json; j=json.load(open('$O/detail.json')); print('flags [$flag]', j['value'], j['repeats']['value_min'], j['repeats']['value_max'], 'kernel us', j['roofline']['avg_kernel_us'])"
done 2>&1 | tee $O/vreg.txt
