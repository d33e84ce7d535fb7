#!/bin/bash
# round 6, eighth call: the cull as a kernel of its own over the list (k_cull_pairs), two pyramid levels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
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
print("cull=$c", j["value"], j["config"].get("culled_pair_frac"), "kernel us", r["avg_kernel_us"], "insts", r.get("insts_valu"), "front", {k:(v.get("avg_us_alone"), v.get("insts_valu")) for k,v in (r.get("front_chain") or {}).items()}, "single", (j.get("roofline_single_frame") or {}).get("frames_per_s"))
PY
done
( time timeout 1500 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -m gpu -q -x -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
