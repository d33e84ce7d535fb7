#!/bin/bash
# round 6: does removing the dead pairs shorten the integrate kernel when nothing runs beside it?  (overlap=0: one stream)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06k
mkdir -p $O
cd /tmp
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-depth-only --no-single-frame --no-pmc --repeats 1"
for ov in 0 1; do for c in 1 0; do
  rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py $Q --tune cull=$c --tune overlap=$ov > $O/kt_ov${ov}_c$c.log 2>&1
  DB=$(find /tmp/kt -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_ov${ov}_cull$c.txt 2>&1; echo "== overlap=$ov cull=$c"; tail -1 $O/kt_ov${ov}_c$c.log | cut -c1-120; sed -n 3,7p $O/kernel_stats_ov${ov}_cull$c.txt | cut -c1-150
done; done
