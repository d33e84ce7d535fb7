#!/bin/bash
# round 6: 4 mm, one frame per launch (a live stream): the next frame's front chain beside the persistent kernel on the low-priority front stream (pipe_overlap 1) against one stream
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zm
mkdir -p $O
cd $R
for rep in 1 2; do for t in "" "--tune pipe_overlap=1" "--tune pipe_overlap=1 --tune front_prio=1"; do
  timeout 600 python bench.py --single-frame --depth-only --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache > $O/b.json 2> $O/b.err $t
  python - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zm/b.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("tune [%s] value %.1f | kernel us %s frac %s" % (sys.argv[1], d["value"], r.get("avg_kernel_us"), r.get("frac")))
PY
done; done 2>&1 | tee $O/runs.txt
