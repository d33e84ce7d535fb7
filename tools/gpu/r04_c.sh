#!/bin/bash
# round 4, third call: COLOR = 2 kernel check (RGB-D tests + bench), marching-cubes leg, front-chain counters, 1 mm overlap experiments
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -k "rgbd or colour or clamp or conformance or digests or schedules" ) > $O/pytest_colour.log 2>&1
grep -E "passed|failed" $O/pytest_colour.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_colour.log | head
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err
( time timeout 900 python bench.py --no-single-frame --no-out-of-cache --no-e2e-rgbd --cpu-frames 60 ) > $O/bench_full.json 2> $O/bench_full.err; tail -3 $O/bench_full.err
for t in "" "front_prio=0" "front_cus=32" "front_cus=64" "front_cus=96" "pipe_wgs=2"; do
  n=$(echo "base$t" | tr '=' '_')
  ( timeout 300 python bench.py --config 1mm --single-frame --steps 64 --warmup 16 --repeats 1 --no-pmc --no-cpu-baseline ${t:+--tune $t} ) > $O/ooc_$n.json 2> $O/ooc_$n.err
done
python - <<'PY'
import json, glob
for f in ("bench_driver", "bench_full"):
    try:
        j = json.loads([l for l in open("gpurun_out/r04c/%s.json" % f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; d = j.get("roofline_depth_only") or {}
        print(f, "value", j["value"], "depth-only", j.get("value_depth_only"), "| rgbd us/launch", r.get("avg_kernel_us"), "frac", r.get("frac"), "insts/vf", (r.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"),
              "| depth us/launch", d.get("avg_kernel_us"), "| parity", (j.get("parity") or {}).get("sha256_equal"), (j.get("parity_depth_only") or {}).get("sha256_equal"))
        print("   front", json.dumps(r.get("front_chain")))
        print("   mc", json.dumps((j.get("end_to_end") or {}).get("marching_cubes")))
    except Exception as e:
        print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r04c/ooc_*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j["roofline"]
        print(f.split("/")[-1], "frames/s", j["value"], "kernel us", r["avg_kernel_us"], "frac", r["frac"], "ceiling", (r.get("pattern_ceiling") or {}).get("rmw_copy_GBs"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
