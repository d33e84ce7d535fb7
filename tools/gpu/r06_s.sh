#!/bin/bash
# round 6: k_jpeg_huff with 12-bit first-level tables in LDS: parity, end to end, kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
for v in ycc rgb; do
  if [ $v = rgb ]; then export SF_JPEG_RGB_IMAGE=1; else unset SF_JPEG_RGB_IMAGE; fi
  ( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 ) > "$O/e2e_$v.log" 2>&1
  echo "e2e $v"; grep "sf_fuse_run:" "$O/e2e_$v.log" | cut -c1-200; tail -1 "$O/e2e_$v.log" | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    j=json.loads(l); f=j.get('fuse') or j
    print({k:f.get(k) for k in ('frames_per_s_end_to_end','seconds','first_run_of_the_process')})
except Exception as e: print(l[-600:])"
done
unset SF_JPEG_RGB_IMAGE
cd /tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 > $O/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_e2e_rgbd.txt 2>&1; head -16 $O/kernel_stats_e2e_rgbd.txt | cut -c1-170
