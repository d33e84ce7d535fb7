#!/bin/bash
# Round 4, seventh GPU call: the device's inflate -- parity tests, then the end-to-end rate of the 5 578-frame scan with it (16 and 4 host
# threads) and without, and the kernel trace of a short run.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run or drop_in" ) > $O/pytest_inflate.log 2>&1
tail -15 $O/pytest_inflate.log
grep -q "failed\|error" $O/pytest_inflate.log && exit 1
( timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --out $O/e2e_5578_gpu_inflate.json ) > $O/e2e_5578_gpu_inflate.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --threads 4 --out $O/e2e_5578_gpu_inflate_4_threads.json ) > $O/e2e_5578_gpu_inflate_4_threads.log 2>&1
( SF_INFLATE_HOST=1 timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --out $O/e2e_5578_host_inflate.json ) > $O/e2e_5578_host_inflate.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g/e2e*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], j["fuse"])
PY
cd /tmp
rm -rf /tmp/kti; SF_RUN_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kti -o kt -- python $R/tools/e2e_bench.py --frames 1200 --fuse-only --out /tmp/e2e_kt.json > $O/kt_inflate.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kti -name "*.db" | head -1) > $O/kt_inflate.txt 2>&1
head -12 $O/kt_inflate.txt | cut -c1-150
grep "sf_fuse_run" $O/kt_inflate.log | tail -2
