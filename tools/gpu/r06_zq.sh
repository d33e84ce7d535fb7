#!/bin/bash
# round 6: the 20-frame call under rocprofv3 --kernel-trace, this tree's library and the session's first commit's: which kernel differs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zq
mkdir -p $O
cd /tmp
for v in new old; do
  lib=""; [ $v = old ] && lib=$R/tools/experiments/libscanfuse_f8bd621.so
  rm -rf /tmp/kt; SCANFUSE_LIBRARY=$lib timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --teardown > $O/kt_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | head -12 | cut -c1-150
  python $R/tools/timeline.py $(find /tmp/kt -name "*.db" | head -1) --skip k_synth --skip at:: > $O/timeline_$v.txt 2>&1
done 2>&1 | tee $O/stats.txt
