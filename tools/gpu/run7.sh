#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_simplify_gpu.py -m gpu -q -s ) > $O/pytest_simplify.log 2>&1
grep -E "passed|failed|decimate 977k|Error|error" $O/pytest_simplify.log | head -20
cd /tmp
rm -rf /tmp/ktc; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktc -o kt -- python $R/tools/e2e_bench.py --frames 2000 --color raw --fuse-only > $O/kt_colour_raw.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/ktc -name "*.db" | head -1) > $O/kt_colour_raw.txt 2>&1
rm -rf /tmp/ktj; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktj -o kt -- python $R/tools/e2e_bench.py --frames 1500 --color jpeg --color-res 1296x968 --fuse-only > $O/kt_colour_jpeg1296.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/ktj -name "*.db" | head -1) > $O/kt_colour_jpeg1296.txt 2>&1
head -12 $O/kt_colour_raw.txt | cut -c1-140; head -12 $O/kt_colour_jpeg1296.txt | cut -c1-140
