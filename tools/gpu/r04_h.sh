#!/bin/bash
# Round 4: the device's inflate, kernels alone (tools/gpu/inflate_bench.py) + its parity test + the end-to-end rate
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run" ) > $O/pytest_inflate.log 2>&1
tail -4 $O/pytest_inflate.log
( timeout 300 python tools/gpu/inflate_bench.py ) > $O/inflate_bench.json 2> $O/inflate_bench.err; cat $O/inflate_bench.json; tail -3 $O/inflate_bench.err
grep -q "failed\|error" $O/pytest_inflate.log && exit 1
( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --out $O/e2e_5578_gpu_inflate.json ) > $O/e2e_5578_gpu_inflate.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04h/e2e*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], j["fuse"])
PY
( timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --threads 4 --out $O/e2e_5578_gpu_inflate_4_threads.json ) > $O/e2e_5578_gpu_inflate_4_threads.log 2>&1
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04h/e2e_5578_gpu_inflate_4_threads.json")); print("4 threads", j["fuse"])
PY
grep "sf_fuse_run" $O/e2e_5578_gpu_inflate.log | tail -2 | cut -c1-250
