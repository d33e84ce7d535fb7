#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run or colour" ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw_gpu.json ) > $O/e2e_colour_raw_gpu.log 2>&1
grep "sf_fuse_run" $O/e2e_colour_raw_gpu.log | tail -2 | cut -c1-300
( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --threads 4 --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
grep "sf_fuse_run" $O/e2e_5578_gpu.log | tail -2 | cut -c1-300
( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 1024 --fuse-only --threads 4 --out $O/e2e_1024_gpu.json ) > $O/e2e_1024_gpu.log 2>&1
grep "sf_fuse_run" $O/e2e_1024_gpu.log | tail -2 | cut -c1-300
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
python - <<'PY'
import json
for n in ("e2e_colour_raw_gpu", "e2e_5578_gpu", "e2e_1024_gpu", "e2e_colour_jpeg_1296"):
    print(n, json.load(open("gpurun_out/r04l/%s.json" % n))["fuse"]["frames_per_s_end_to_end"])
PY
