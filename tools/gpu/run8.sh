#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log
( timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color jpeg --fuse-only --out $O/e2e_colour_jpeg.json ) > $O/e2e_colour_jpeg.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu-decimate ) > $O/bench_scans_gpudec.json 2> $O/bench_scans_gpudec.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2h/*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0]) if f.endswith("json") and "bench" in f else json.load(open(f))
        if "fuse" in j: print(f.split("/")[-1], j["fuse"]["frames_per_s_end_to_end"])
        else:
            r = j.get("roofline") or {}; c = j.get("roofline_colour") or {}; s1 = j.get("roofline_single_frame") or {}
            print(f.split("/")[-1], j["value"], j["unit"], "kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "| colour", c.get("frames_per_s"), c.get("avg_kernel_us"), "| single", s1.get("frames_per_s"), s1.get("frac"), "| idle", j.get("gpu_idle_pct"), j.get("host_stage_s_mean_rank0"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/bench_scans_gpudec.err
