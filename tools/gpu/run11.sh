#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2k
mkdir -p $O
cd /tmp
rm -rf /tmp/kt_dec
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_dec -o dec -- python $R/tools/e2e_bench.py --frames 5578 --gpu-decimate --out $O/e2e_5578_gpudec.json ) > $O/kt_dec.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt_dec -name "*.db" | head -1) > $O/kt_dec.txt 2>&1
head -40 $O/kt_dec.txt | cut -c1-220
