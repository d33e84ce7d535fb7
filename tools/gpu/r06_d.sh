#!/bin/bash
# round 6, fourth call: compaction with 16 waves per workgroup, allocation group sweep (round balance), end to end RGB-D: packed segment copies and side-stream
# priority A/B, the whole default bench (5578-frame RGB-D leg)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06d
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -m gpu -q -x -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --repeats 3"
( SF_BENCH_DETAIL=$O/detail.json timeout 600 python bench.py $Q ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.load(open("$O/detail.json")); r=j["roofline"]
print("default", j["value"], "kernel us", r["avg_kernel_us"], "front", {k:(v.get("avg_us_alone"), v.get("insts_valu")) for k,v in (r.get("front_chain") or {}).items()})
PY
for g in 5 6 7 8 11 32; do
  ( SF_BENCH_DETAIL=$O/detail_ag$g.json timeout 300 python bench.py $Q --no-pmc --tune alloc_group=$g ) > $O/bench_ag$g.json 2> $O/bench_ag$g.err
  python -c "
import json; j=json.load(open('$O/detail_ag$g.json')); print('alloc_group $g', j['value'], j['repeats']['value_min'], j['repeats']['value_max'], 'kernel us', j['roofline']['avg_kernel_us'])"
done
# end to end RGB-D, side streams at high priority (this build) ...
( timeout 600 python tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 ) > $O/e2e_prio1.log 2>&1; tail -1 $O/e2e_prio1.log | cut -c1-700
( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 ) > $O/e2e_prio1_timing.log 2>&1; grep "sf_fuse_run:" $O/e2e_prio1_timing.log | cut -c1-400
# ... and at the default priority (pipeline.hip rebuilt with -DSF_SIDE_PRIO=0)
touch scannet_amd/csrc/pipeline.hip
( SCANFUSE_BUILD_FLAGS=-DSF_SIDE_PRIO=0 timeout 900 python -c "from scannet_amd import build; build.build()" ) > $O/rebuild0.log 2>&1; tail -2 $O/rebuild0.log
( timeout 600 python tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 ) > $O/e2e_prio0.log 2>&1; tail -1 $O/e2e_prio0.log | cut -c1-700
touch scannet_amd/csrc/pipeline.hip
( timeout 900 python -c "from scannet_amd import build; build.build()" ) > $O/rebuild1.log 2>&1; tail -2 $O/rebuild1.log
# the whole default command
( time SF_BENCH_DETAIL=$O/detail_default.json timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 3600 $O/bench_default.json; tail -3 $O/bench_default.err
