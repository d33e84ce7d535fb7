#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-tl}
mkdir -p $O
cd /tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --repeats 1 --teardown > $O/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/timeline.py $DB > $O/timeline.txt 2>&1
python $R/tools/rocpd_summary.py $DB > $O/kt.txt 2>&1
cd $R
for t in "alloc_group=4" "alloc_group=8" "xcd_walk=0" "ramp=16"; do
  timeout 200 python bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --repeats 3 --tune $t > $O/tune_$t.json 2> $O/tune_$t.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/tune_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0]); print(f.split("/")[-1], j["value"], j["roofline"]["avg_kernel_us"])
    except Exception as e: print(f,"ERR",e)
PY
cat $O/timeline.txt | head -75
