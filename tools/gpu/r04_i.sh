#!/bin/bash
# Round 4: kernel timeline of sf_fuse_run with the device's inflate
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
cd /tmp
rm -rf /tmp/kti; SF_RUN_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kti -o kt -- python $R/tools/e2e_bench.py --frames 2400 --fuse-only --out /tmp/e2e_kt.json > $O/kt_inflate.log 2>&1
DB=$(find /tmp/kti -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/kt_inflate.txt 2>&1
python $R/tools/timeline.py $DB 2640 130 > $O/timeline_inflate.txt 2>&1
grep "sf_fuse_run" $O/kt_inflate.log | tail -2
cat $O/timeline_inflate.txt | cut -c1-120
