#!/bin/bash
# round 6: the 20-frame call (the driver's command) with the library of every commit of this session, one box, twice each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zr
mkdir -p $O
cd $R
for rep in 1 2; do for c in f8bd621 tree; do
  lib=$R/tools/experiments/libscanfuse_$c.so; [ $c = tree ] && lib=""
  SCANFUSE_LIBRARY=$lib timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-out-of-cache --no-single-frame --no-e2e > $O/b.json 2> $O/b.err
  python - "$c" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06zr/b.json").read().strip().splitlines()[-1])
print("[%s] value %.1f depth-only %s kernel us %s" % (sys.argv[1], d["value"], d.get("value_depth_only"), d["roofline"].get("avg_kernel_us")))
PY
done; done 2>&1 | tee $O/runs.txt
timeout 1200 python -m pytest tests/test_gpu_tsdf.py -q -x -m gpu -k "presence_cache or allocation_kernels or furnished or one_mm or garbage or reset" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
