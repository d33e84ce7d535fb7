#!/bin/bash
# round 6: brick-local hash home: 1 mm probe, parity, headline A/B is the bench line itself
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
timeout 600 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -3
( time timeout 1500 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -m gpu -q -x ) > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest.log | head
Q="--no-cpu-baseline --no-e2e --no-depth-only --repeats 3"
( SF_BENCH_DETAIL=$O/detail.json timeout 900 python bench.py $Q ) > $O/bench.json 2> $O/bench.err
python -c "
import json; j=json.load(open('$O/detail.json')); r=j['roofline']; ro=j.get('roofline_out_of_cache') or {}
print('value', j['value'], 'kernel us', r['avg_kernel_us'], 'front', {k:v.get('avg_us_alone') for k,v in (r.get('front_chain') or {}).items()}, 'single', (j.get('roofline_single_frame') or {}).get('frames_per_s'))
print('ooc frac', ro.get('frac'), 'us', ro.get('avg_kernel_us'), 'fps', ro.get('frames_per_s'), 'alone', (ro.get('kernel_alone') or {}).get('frac'), (ro.get('kernel_alone') or {}).get('frames_per_s'), 'batched', ro.get('batched_frames_per_s'))"
cd /tmp; rm -rf /tmp/kt; SF_PROBE_ONLY_BATCH1=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep -E "k_alloc<6|k_compactify|k_integrate_pipe" | cut -c1-140
