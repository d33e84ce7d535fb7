#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2m
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_simplify_gpu.py -m gpu -q -s ) > $O/pytest_simplify.log 2>&1
grep -E "passed|failed|flat 977k" $O/pytest_simplify.log | tail -3; grep -E "^FAILED|^E " $O/pytest_simplify.log | head
cd /tmp; rm -rf /tmp/kt_dec
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_dec -o dec -- python $R/tools/e2e_bench.py --frames 5578 --gpu-decimate --out $O/e2e_5578_gpudec.json ) > $O/kt_dec.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt_dec -name "*.db" | head -1) > $O/kt_dec.txt 2>&1
head -12 $O/kt_dec.txt | cut -c1-160
cd $R
python -c "
import json; j=json.load(open('$O/e2e_5578_gpudec.json')); print({k:j[k] for k in j if k.startswith('decimate') or k.startswith('seg')})"
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu-decimate ) > $O/bench_scans_gpudec.json 2> $O/bench_scans_gpudec.err
tail -3 $O/bench_scans_gpudec.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r2m/bench_scans_gpudec.json").read().splitlines() if l.startswith("{")][0])
print(j["value"], j["unit"], "idle", j["gpu_idle_pct"], "busy", j["gpu_busy_s_sum"], "host", j["host_stage_s_mean_rank0"], j["host_stage_parts_s_mean_rank0"])
PY
