#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2n
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_meshclean_gpu.py -m gpu -q -s -x ) > $O/pytest_clean.log 2>&1
grep -E "passed|failed|clean [0-9]+ faces" $O/pytest_clean.log | tail -3; grep -E "^FAILED|^E  " $O/pytest_clean.log | head -20
( timeout 900 python tools/e2e_bench.py --frames 5578 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
python -c "
import json; j=json.load(open('$O/e2e_5578_gpu.json')); print({k:j[k] for k in j if k.endswith('_s') or k.startswith('seg')})"
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu ) > $O/bench_scans_gpu.json 2> $O/bench_scans_gpu.err
tail -3 $O/bench_scans_gpu.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r2n/bench_scans_gpu.json").read().splitlines() if l.startswith("{")][0])
print(j["value"], j["unit"], "idle", j["gpu_idle_pct"], "busy", j["gpu_busy_s_sum"], "host", j["host_stage_s_mean_rank0"], j["host_stage_parts_s_mean_rank0"])
PY
( timeout 900 python bench.py --config 1mm --no-cpu-baseline --no-pmc ) > $O/bench_1mm.json 2> $O/bench_1mm.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r2n/bench_1mm.json").read().splitlines() if l.startswith("{")][0])
s1 = j["roofline_single_frame"]; print(j["value"], s1["frac"], s1["avg_kernel_us"], s1.get("kernel_alone"))
PY
