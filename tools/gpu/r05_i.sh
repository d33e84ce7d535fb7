#!/bin/bash
# Round 5, call I: where an RGB-D scan's end-to-end time goes with the device's entropy decoder (1296x968 JPEG colour of ~200 KB, 4 host threads):
# the run's own phase clock (SF_RUN_TIMING) and the kernel trace.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
cd /tmp
export SF_JPEG_GPU_HUFFMAN=1 SF_RUN_TIMING=1
( timeout 600 python $R/tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 --out $O/e2e.json ) > $O/e2e.log 2>&1; grep -E "sf_fuse_run|frames_per_s" $O/e2e.log | cut -c1-400
rm -rf /tmp/kt_i; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_i -o kt -- python $R/tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt_i -name "*.db" | head -1) 2>&1 | head -16 | cut -c1-150 | tee $O/kt.txt
python $R/tools/timeline.py $(find /tmp/kt_i -name "*.db" | head -1) -260 90 --skip k_synth > $O/timeline.txt 2>&1; head -95 $O/timeline.txt
