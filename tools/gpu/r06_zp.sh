#!/bin/bash
# round 6: the 20-frame call (the driver's command): this tree with the presence cache on / off against the session's first commit, one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zp
mkdir -p $O
cd $R
for rep in 1 2 3; do for v in "new" "old"; do
  lib=""; t=""
  case "$v" in old) lib=$R/tools/experiments/libscanfuse_f8bd621.so;; "new --tune brick_cache=0") t="--tune brick_cache=0";; esac
  SCANFUSE_LIBRARY=$lib timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-out-of-cache --no-single-frame --no-e2e $t > $O/b.json 2> $O/b.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zp/b.json").read().strip().splitlines()[-1])
print("[%s] value %.1f depth-only %s kernel us %s" % (sys.argv[1], d["value"], d.get("value_depth_only"), d["roofline"].get("avg_kernel_us")))
PY
done; done 2>&1 | tee $O/runs.txt
