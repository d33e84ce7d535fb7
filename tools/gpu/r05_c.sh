#!/bin/bash
# (run while the packed pairs were the default build: -DSF_SCALAR_PAIRS selected the plain pairs then; since then plain pairs are the default and -DSF_PACKED_PAIRS selects the round-4 kernels)
# Round 5, call C: packed pairs (v_pk_*_f32) against plain fp32 pairs in the integrate kernels -- the A/B the VALU issue table asks for.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
QUICK="--no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-pmc --repeats 3"
( timeout 600 python bench.py $QUICK ) > $O/bench_packed.json 2> $O/bench_packed.err
export SCANFUSE_BUILD_FLAGS="-DSF_SCALAR_PAIRS -fno-slp-vectorize"
( time python -c "from scannet_amd import build; build.build(force=True)" ) > $O/build_scalar.log 2>&1; tail -3 $O/build_scalar.log
( timeout 600 python bench.py $QUICK ) > $O/bench_scalar.json 2> $O/bench_scalar.err
( timeout 900 python -m pytest tests/test_gpu_tsdf.py -q -x -k "rgbd or batched or clamp or smoke or conformance" ) > $O/pytest_scalar.log 2>&1; tail -3 $O/pytest_scalar.log
python - <<'PY'
import json
for v in ("packed", "scalar"):
    try:
        j = json.loads([l for l in open("gpurun_out/r05c/bench_%s.json" % v).read().splitlines() if l.startswith("{")][0])
        r = j["roofline"]; d = j.get("roofline_depth_only") or {}
        print(v, "value", j["value"], "depth only", j.get("value_depth_only"), "kernel us", r.get("avg_kernel_us"), d.get("avg_kernel_us"), "parity", j.get("parity", {}).get("sha256_equal"))
    except Exception as ex:
        print(v, "failed", ex)
PY
tail -2 $O/bench_scalar.err
