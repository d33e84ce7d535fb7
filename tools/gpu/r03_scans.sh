#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence3
mkdir -p $O
cd $R
( timeout 120 python -m pytest tests/test_synth.py -m gpu -q 2>&1 | tail -2 )
for hs in none gpu; do
  ( timeout 300 python bench.py --config scans --steps 12 --host-stage $hs ) > $O/bench_scans_$hs.json 2> $O/bench_scans_$hs.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_scans_$hs.json") if l.startswith("{")][0]); print("$hs", j["value"], j["unit"], "idle", j["gpu_idle_pct"], "host", j["host_stage_s_mean_rank0"], "frames/s", j["frames_per_s"])
except Exception as e: print("$hs ERR", e, open("$O/bench_scans_$hs.err").read()[-300:])
PY
done
