#!/bin/bash
# round 6: the whole default bench command (its 1 mm legs run before the end-to-end legs in the same process) with this tree's library and the session's first commit's, one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zu
mkdir -p $O
cd $R
for rep in 1 2; do for c in f8bd621 tree; do
  lib=$R/tools/experiments/libscanfuse_$c.so; [ $c = tree ] && lib=""
  SCANFUSE_LIBRARY=$lib timeout 900 python bench.py --no-cpu-baseline --no-pmc > $O/b.json 2> $O/b.err
  python - "$c" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zu/b.json").read().strip().splitlines()[-1]); e = d.get("end_to_end") or {}; r = d["roofline"]
print("[%s] value %.1f | ooc %s | single %s | e2e rgbd first %s best %s | e2e depth-only %s" % (sys.argv[1], d["value"], (r.get("hbm_out_of_cache") or {}).get("frac"), (d.get("roofline_single_frame") or {}).get("frames_per_s"),
      e.get("frames_per_s"), e.get("frames_per_s_best"), (e.get("depth_only") or {}).get("frames_per_s")))
PY
done; done 2>&1 | tee $O/runs.txt
