#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run or colour" ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
python - <<'PY'
import json
for n in ("e2e_colour_raw", "e2e_colour_jpeg_1296"):
    print(n, json.load(open("gpurun_out/r04k/%s.json" % n))["fuse"]["frames_per_s_end_to_end"])
PY
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04k/bench_4mm.json").read().splitlines() if l.startswith("{")][0])
r = j["roofline"]
print(j["value"], j.get("value_depth_only"), r["avg_kernel_us"], r["frac"], (r.get("hbm_out_of_cache") or {}).get("frac"))
print("e2e", {k: v for k, v in (j.get("end_to_end") or {}).items() if k in ("frames", "frames_per_s", "seconds", "decode_threads", "decode_ms_per_frame_per_thread")})
print("e2e rgbd", {k: v for k, v in (j.get("end_to_end_rgbd") or {}).items() if k in ("frames", "frames_per_s", "seconds", "decode_threads", "decode_ms_per_frame_per_thread")})
print(j.get("parity", {}).get("sha256_equal"))
PY
tail -3 $O/bench_4mm.err
