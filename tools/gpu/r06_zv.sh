#!/bin/bash
# round 6: the second front stream at the main stream's priority or at the lowest: the 1 mm rates, then the whole default command's end-to-end legs behind its 1 mm legs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zv
mkdir -p $O
cd $R
for rep in 1 2; do for t in "--tune front_lo_lowest=0" "--tune front_lo_lowest=1" "--tune front_prio=1"; do
  timeout 600 python bench.py --config 1mm --no-cpu-baseline --no-pmc $t > $O/b.json 2> $O/b.err
  python - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zv/b.json").read().strip().splitlines()[-1])
print("1 mm [%s] value %.1f single_frame %s" % (sys.argv[1], d["value"], d.get("roofline_single_frame")))
PY
done; done 2>&1 | tee $O/runs.txt
for rep in 1 2; do for t in "--tune front_lo_lowest=0" "--tune front_lo_lowest=1"; do
  timeout 900 python bench.py --no-cpu-baseline --no-pmc $t > $O/b.json 2> $O/b.err
  python - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zv/b.json").read().strip().splitlines()[-1]); e = d.get("end_to_end") or {}; r = d["roofline"]
print("[%s] value %.1f | ooc %s | e2e rgbd first %s best %s | e2e depth-only %s" % (sys.argv[1], d["value"], (r.get("hbm_out_of_cache") or {}).get("frac"), e.get("frames_per_s"), e.get("frames_per_s_best"), (e.get("depth_only") or {}).get("frames_per_s")))
PY
done; done 2>&1 | tee -a $O/runs.txt
