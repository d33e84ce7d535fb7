#!/bin/bash
# round 6: 1 mm, one frame per launch: is the rate bistable, and does the front stream's priority decide it?  (bench.py --config 1mm, three runs each)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zk
mkdir -p $O
cd $R
for rep in 1 2 3; do for t in "" "--tune front_prio=0" "--tune pipe_wgs=1"; do
  timeout 600 python bench.py --config 1mm --no-cpu-baseline --no-pmc $t > $O/b.json 2> $O/b.err
  python - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zk/b.json").read().strip().splitlines()[-1])
print("tune [%s] value %.1f single_frame %s" % (sys.argv[1], d["value"], d.get("roofline_single_frame")))
PY
done; done 2>&1 | tee $O/runs.txt
