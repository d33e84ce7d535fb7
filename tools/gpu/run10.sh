#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_simplify_gpu.py -m gpu -q -s ) > $O/pytest_simplify.log 2>&1
grep -E "passed|failed|flat 977k" $O/pytest_simplify.log | tail -3; grep -E "^FAILED" $O/pytest_simplify.log
( timeout 900 python tools/e2e_bench.py --frames 5578 --gpu-decimate --out $O/e2e_5578_gpudec.json ) > $O/e2e_5578_gpudec.log 2>&1
tail -2 $O/e2e_5578_gpudec.log | cut -c1-1500
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu-decimate ) > $O/bench_scans_gpudec.json 2> $O/bench_scans_gpudec.err
tail -3 $O/bench_scans_gpudec.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r2j/bench_scans_gpudec.json").read().splitlines() if l.startswith("{")][0])
print(j["value"], j["unit"], "idle", j["gpu_idle_pct"], "busy", j["gpu_busy_s_sum"], "host", j["host_stage_s_mean_rank0"], j["host_stage_parts_s_mean_rank0"])
PY
