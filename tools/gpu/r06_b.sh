#!/bin/bash
# round 6, second call: parity of the x-row integrate layout and the (entry, frame)-lane compaction, A/B of both, the decimation comparison
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06b
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -m gpu -q -x -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --repeats 3"
for t in xrow=1 xrow=0; do
  ( SF_BENCH_DETAIL=$O/detail_$t.json timeout 600 python bench.py $Q --tune $t ) > $O/bench_$t.json 2> $O/bench_$t.err
  python - <<PY
import json
j=json.load(open("$O/detail_$t.json")); r=j["roofline"]
print("$t", j["value"], "kernel us", r["avg_kernel_us"], "valu2", r.get("valu_frac_2cycle"), "ta", r.get("ta_busy"), "tcp/clk", r.get("tcp_tag_lookups_per_cu_clk"), "tcp/gather", (r.get("mem_pipe") or {}).get("tcp_tag_lookups_per_gather"), "front", {k:v.get("avg_us_alone") for k,v in (r.get("front_chain") or {}).items()})
PY
done
( SF_BENCH_DETAIL=$O/detail_driver.json timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --no-pmc ) > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-300 $O/bench_driver.json
( time timeout 900 python tools/decimate_compare.py --out $O/decimate_compare.json ) > $O/decimate.log 2>&1; tail -5 $O/decimate.log | cut -c1-300
