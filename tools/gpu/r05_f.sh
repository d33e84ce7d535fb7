#!/bin/bash
# (run while the packed pairs were the default build: -DSF_SCALAR_PAIRS selected the plain pairs then; since then plain pairs are the default and -DSF_PACKED_PAIRS selects the round-4 kernels)
# Round 5, call F: occupancy of the integrate kernel -- the tile fused in halves (fewer live registers, more waves per SIMD), plain and packed pairs.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
QUICK="--no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-pmc --repeats 3"
for v in "$@"; do
  name=${v%%:*}; rest="${v#*:}"; export SCANFUSE_BUILD_FLAGS="${rest%%:*}"; extra=""; [ "$rest" != "${rest#*:}" ] && extra="${rest#*:}"   # name:build flags[:bench flags]
  python -c "from scannet_amd import build; build.build(force=True)" > $O/build_$name.log 2>&1
  ( timeout 600 python bench.py $QUICK $extra ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $name <<'PY'
import json, sys
v = sys.argv[1]
try:
    j = json.loads([l for l in open("gpurun_out/r05f/bench_%s.json" % v).read().splitlines() if l.startswith("{")][0])
    r = j["roofline"]; d = j.get("roofline_depth_only") or {}
    print(v, "value", j["value"], "depth only", j.get("value_depth_only"), "kernel us", r.get("avg_kernel_us"), d.get("avg_kernel_us"))
except Exception as ex:
    print(v, "failed", ex)
PY
done
( timeout 900 python -m pytest tests/test_gpu_tsdf.py -q -x -k "rgbd or batched or clamp or smoke or conformance" ) > $O/pytest_last_variant.log 2>&1; tail -2 $O/pytest_last_variant.log
