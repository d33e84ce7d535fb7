#!/bin/bash
# round 6: is the end-to-end first run (and the 20-step call) slower with the presence cache, or was it the box?  The driver's command, cache on / off, twice each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zn
mkdir -p $O
cd $R
for rep in 1 2; do for t in "" "--tune brick_cache=0"; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-out-of-cache --no-single-frame $t > $O/b.json 2> $O/b.err
  python - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zn/b.json").read().strip().splitlines()[-1]); e = d.get("end_to_end") or {}
print("tune [%s] value %.1f depth-only %s | e2e rgbd first %s best %s | e2e depth-only %s" % (sys.argv[1], d["value"], d.get("value_depth_only"), e.get("frames_per_s"), e.get("frames_per_s_best"), (e.get("depth_only") or {}).get("frames_per_s")))
PY
done; done 2>&1 | tee $O/runs.txt
