#!/bin/bash
# round 6: 1 mm voxels, one frame per launch (tiles 25 x the Infinity Cache): the front chain on CUs of its own beside k_integrate_pipe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-e2e --no-depth-only --no-single-frame --no-pmc --repeats 1"
for fc in 0 32 64 96 128; do
  ( SF_BENCH_DETAIL=$O/detail_fc$fc.json timeout 600 python bench.py $Q --tune front_cus=$fc ) > $O/bench_fc$fc.json 2> $O/bench_fc$fc.err
  python -c "
import json; j=json.load(open('$O/detail_fc$fc.json')); ro=j.get('roofline_out_of_cache') or {}
print('front_cus=$fc', 'value', j['value'], 'ooc frac', ro.get('frac'), 'us', ro.get('avg_kernel_us'), 'fps', ro.get('frames_per_s'), 'alone', (ro.get('kernel_alone') or {}).get('frac'), (ro.get('kernel_alone') or {}).get('frames_per_s'), 'batched', ro.get('batched_frames_per_s'))"
done
