#!/bin/bash
# round 4, fifth call: the per-axis allocation walk -- every test that compares block sets, then the two bench commands
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -x ) > $O/pytest_tsdf.log 2>&1
grep -E "passed|failed" $O/pytest_tsdf.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_tsdf.log | head
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
X="--no-cpu-baseline --no-single-frame --no-out-of-cache --no-e2e"
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 $X ) > $O/bench_driver.json 2> $O/bench_driver.err
( timeout 600 python bench.py $X ) > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
for f in ("bench_driver", "bench_full"):
    try:
        j = json.loads([l for l in open("gpurun_out/r04e/%s.json" % f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}
        print(f, "value", j["value"], "depth-only", j.get("value_depth_only"), "| rgbd us/launch", r.get("avg_kernel_us"), "frac", r.get("frac"))
        print("   front", json.dumps((r.get("front_chain") or {}).get("k_alloc_ray")))
    except Exception as e:
        print(f, "ERR", e)
PY
