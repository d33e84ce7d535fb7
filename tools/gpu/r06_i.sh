#!/bin/bash
# round 6: kernel traces of the default stream with and without the dead-pair cull (where does the time go?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i
mkdir -p $O
cd /tmp
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-depth-only --no-single-frame --no-pmc --repeats 1"
for c in 1 0; do
  rm -rf /tmp/kt$c; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt$c -o kt -- python $R/bench.py $Q --tune cull=$c > $O/kt$c.log 2>&1
  DB=$(find /tmp/kt$c -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_cull$c.txt 2>&1; echo "== cull=$c"; tail -1 $O/kt$c.log | cut -c1-200; head -12 $O/kernel_stats_cull$c.txt | cut -c1-170
done
