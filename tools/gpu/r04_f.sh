#!/bin/bash
# Round 4, sixth GPU call: the device's JPEG entropy decoder -- parity tests, the RGB-D end-to-end rate with it and without, its kernel time;
# then the kernel trace of the driver's command again with the launches of bench.py's event sample singled out.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "jpeg or colour" ) > $O/pytest_jpeg.log 2>&1
tail -5 $O/pytest_jpeg.log
( SF_JPEG_GPU_HUFFMAN=1 timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296_host_huffman.json ) > $O/e2e_colour_jpeg_1296_host_huffman.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f/e2e*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], j["fuse"])
PY
cd /tmp
rm -rf /tmp/ktj; SF_JPEG_GPU_HUFFMAN=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktj -o kt -- python $R/tools/e2e_bench.py --frames 600 --color jpeg --color-res 1296x968 --fuse-only --out /tmp/e2e_kt.json > $O/kt_jpeg.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/ktj -name "*.db" | head -1) > $O/kt_jpeg.txt 2>&1
head -14 $O/kt_jpeg.txt | cut -c1-150
rm -rf /tmp/ktd; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktd -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-e2e --no-single-frame --no-out-of-cache --no-depth-only --teardown > $O/kt_driver.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/ktd -name "*.db" | head -1) --tail "k_integrate<1, 2" 14 --tail "k_integrate<1, 2" 28 > $O/kt_driver.txt 2>&1
python $R/tools/timeline.py $(find /tmp/ktd -name "*.db" | head -1) > $O/timeline.txt 2>&1
grep -h "last\|k_integrate" $O/kt_driver.txt | cut -c1-170; tail -3 $O/timeline.txt; tail -2 $O/kt_driver.log | cut -c1-600
