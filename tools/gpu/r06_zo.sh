#!/bin/bash
# round 6: the library of the session's first commit (f8bd621) against this tree's on ONE box: the driver's command with the end-to-end legs, alternating, twice each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zo
mkdir -p $O
cd $R
for rep in 1 2; do for lib in "" "$R/tools/experiments/libscanfuse_f8bd621.so"; do
  SCANFUSE_LIBRARY=$lib timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-out-of-cache --no-single-frame > $O/b.json 2> $O/b.err
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zo/b.json").read().strip().splitlines()[-1]); e = d.get("end_to_end") or {}
print("lib [%s] value %.1f depth-only %s | e2e rgbd first %s best %s | e2e depth-only %s" % (sys.argv[1][-22:], d["value"], d.get("value_depth_only"), e.get("frames_per_s"), e.get("frames_per_s_best"), (e.get("depth_only") or {}).get("frames_per_s")))
PY
done; done 2>&1 | tee $O/runs.txt
