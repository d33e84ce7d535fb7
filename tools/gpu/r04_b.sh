#!/bin/bash
# round 4, second call: suite (conformance digests, RGB-D parity, pair decode in the pipeline), bench lines, the whole-scan chain with marching-cubes phases
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err; tail -3 $O/bench_full.err
( timeout 900 python tools/e2e_bench.py --frames 5578 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1; tail -2 $O/e2e_5578_gpu.log | cut -c1-1500
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1; tail -1 $O/e2e_colour_jpeg_1296.log | cut -c1-600
python - <<'PY'
import json
for f in ("bench_driver", "bench_full"):
    try:
        j = json.loads([l for l in open("gpurun_out/r04b/%s.json" % f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; d = j.get("roofline_depth_only") or {}
        print(f, "value", j["value"], "depth-only", j.get("value_depth_only"), "| rgbd us/launch", r.get("avg_kernel_us"), "frac", r.get("frac"), "insts/vf", (r.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"),
              "| depth us/launch", d.get("avg_kernel_us"), "frac", d.get("frac"), (d.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"),
              "| e2e", (j.get("end_to_end") or {}).get("frames_per_s"), (j.get("end_to_end") or {}).get("decode_ms_per_frame_per_thread"), "e2e rgbd", (j.get("end_to_end_rgbd") or {}).get("frames_per_s"),
              "| parity", (j.get("parity") or {}).get("sha256_equal"), (j.get("parity_depth_only") or {}).get("sha256_equal"),
              "| ooc", (r.get("hbm_out_of_cache") or {}).get("frac"), (r.get("hbm_out_of_cache") or {}).get("kernel_alone_frac"), (j.get("roofline_out_of_cache") or {}).get("frames_per_s"), ((j.get("roofline_out_of_cache") or {}).get("kernel_alone") or {}).get("frames_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
