#!/bin/bash
# Round 5, call H: the bench line of the build at HEAD (end_to_end on the reference writer's file; end_to_end_rgbd with ~200 KB pictures and the device /
# host entropy-decoding alternatives); the GPU tests that touch the pipeline.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
( time timeout 900 python bench.py --no-cpu-baseline --no-out-of-cache --no-single-frame --no-pmc --repeats 3 ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/r05h/bench.json").read().splitlines() if l.startswith("{")][0])
    e = j["end_to_end"]; g = j.get("end_to_end_rgbd") or {}
    print("value", j["value"], j.get("value_depth_only"), "kernel us", j["roofline"]["avg_kernel_us"])
    print("e2e", e["writer"], e["frames_per_s"], e["frames_per_s_first_and_second_run"], "other", (e["other_writer"] or {}).get("frames_per_s"), {k: v for k, v in (e["inflate_kernels"] or {}).items() if k != "what"})
    print("rgbd", g.get("frames_per_s"), g.get("frames_per_s_first_and_second_run"), g.get("jpeg_bytes_per_picture"), g.get("decode_threads"), g.get("jpeg_entropy_on_device"))
    for k, v in (g.get("alternatives") or {}).items():
        print("   ", k, v)
except Exception as ex:
    print("bench parse failed", ex)
PY
tail -3 $O/bench.err
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "colour or jpeg or fuse_run" ) > $O/pytest_pipeline.log 2>&1; tail -4 $O/pytest_pipeline.log
