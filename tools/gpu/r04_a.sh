#!/bin/bash
# round 4, first call: the whole -m gpu suite (new: full-size RGB-D parity, bare --gpus 2, direct-path counter), smoke, the driver's bench command and the default one
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err; tail -3 $O/bench_full.err
python - <<'PY'
import json
for f in ("bench_driver", "bench_full"):
    try:
        j = json.loads([l for l in open("gpurun_out/r04a/%s.json" % f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; d = j.get("roofline_depth_only") or {}
        print(f, "value", j["value"], "depth-only", j.get("value_depth_only"), "| rgbd us/launch", r.get("avg_kernel_us"), "frac", r.get("frac"), "insts/vf", (r.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"),
              "| depth us/launch", d.get("avg_kernel_us"), "frac", d.get("frac"), (d.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"),
              "| e2e", (j.get("end_to_end") or {}).get("frames_per_s"), "e2e rgbd", (j.get("end_to_end_rgbd") or {}).get("frames_per_s"),
              "| parity", (j.get("parity") or {}).get("sha256_equal"), (j.get("parity_depth_only") or {}).get("sha256_equal"),
              "| ooc", (r.get("hbm_out_of_cache") or {}), "| 1f", (j.get("roofline_single_frame") or {}).get("frames_per_s"), ((j.get("roofline_single_frame") or {}).get("rgbd_one_frame_per_launch") or {}))
    except Exception as e:
        print(f, "ERR", e)
PY
