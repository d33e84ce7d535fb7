#!/bin/bash
# round 6: the dead-pair cull inside the integrate kernel (lane = frame of the block's mask): parity, A/B, kernel traces
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06j
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -x -rsxX -s -k "dead_pair or furnished or room_stream or batched_pass or rgbd_baseline or one_frame or one_mm_voxels" ) > $O/pytest_first.log 2>&1
grep -E "passed|failed|culled pairs" $O/pytest_first.log | tail -4; grep -E "^FAILED|^ERROR|Error" $O/pytest_first.log | head
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-depth-only --repeats 3"
for c in 1 0; do
  ( SF_BENCH_DETAIL=$O/detail_cull$c.json timeout 600 python bench.py $Q --tune cull=$c ) > $O/bench_cull$c.json 2> $O/bench_cull$c.err
  python - <<PY
import json
j=json.load(open("$O/detail_cull$c.json")); r=j["roofline"]
print("cull=$c", j["value"], j["repeats"]["value_min"], j["repeats"]["value_max"], j["config"].get("culled_pair_frac"), "kernel us", r["avg_kernel_us"], "front", {k:(v.get("avg_us_alone"), v.get("insts_valu")) for k,v in (r.get("front_chain") or {}).items()}, "single", (j.get("roofline_single_frame") or {}).get("frames_per_s"))
PY
done
cd /tmp
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-depth-only --no-single-frame --no-pmc --repeats 1"
for c in 1 0; do
  rm -rf /tmp/kt$c; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt$c -o kt -- python $R/bench.py $Q --tune cull=$c > $O/kt$c.log 2>&1
  DB=$(find /tmp/kt$c -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_cull$c.txt 2>&1; echo "== cull=$c"; head -7 $O/kernel_stats_cull$c.txt | cut -c1-170
done
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -m gpu -q -x -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
