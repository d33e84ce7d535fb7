#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_simplify_gpu.py -m gpu -q -s ) > $O/pytest_simplify.log 2>&1
grep -E "passed|failed|flat 977k" $O/pytest_simplify.log | tail -3; grep -E "^FAILED" $O/pytest_simplify.log
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu-decimate ) > $O/bench_scans_gpudec.json 2> $O/bench_scans_gpudec.err
tail -3 $O/bench_scans_gpudec.err; cat $O/bench_scans_gpudec.json | head -c 2500; echo
( timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
tail -3 $O/bench_4mm.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r2i/bench_4mm.json").read().splitlines() if l.startswith("{")][0])
r = j.get("roofline") or {}; c = j.get("roofline_colour") or {}; s1 = j.get("roofline_single_frame") or {}
print(j["value"], j["unit"], "kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "| colour", c, "| single", s1.get("frames_per_s"), s1.get("frac"))
PY
