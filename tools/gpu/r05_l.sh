#!/bin/bash
# Round 5, call L: the same RGB-D file fused five times in one process, the run's phase clock on (what made the bench's `alternatives` runs slow?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05l
mkdir -p $O
cd /tmp
( timeout 600 python $R/tools/e2e_bench.py --frames 1024 --color jpeg --color-res 1296x968 --fuse-only --threads 4 ) > $O/e2e.log 2>&1; tail -1 $O/e2e.log | cut -c1-300
SF_RUN_TIMING=1 python - <<'PY' 2>&1 | cut -c1-420
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from scannet_amd import fusion, sens, synth
W, H, cw, ch = 640, 480, 1296, 968
fx, fy, mx, my = synth.intrinsics(W, H)
gp = fusion.default_params(depth_width=W, depth_height=H, fx=fx, fy=fy, mx=mx, my=my)
gp.color_width, gp.color_height = cw, ch
gp.cfx, gp.cfy, gp.cmx, gp.cmy = synth.intrinsics(cw, ch)
sd = sens.SensorData("/tmp/sf_e2e/scene_e2e.sens")
for i, (env, nt) in enumerate(((None, 4), (None, 4), ("SF_JPEG_GPU_HUFFMAN", 4), ("SF_JPEG_GPU_HUFFMAN", 2), ("SF_JPEG_HOST_HUFFMAN", 0), (None, 4), (None, 4))):
    for k in ("SF_JPEG_GPU_HUFFMAN", "SF_JPEG_HOST_HUFFMAN"):
        os.environ.pop(k, None)
    if env:
        os.environ[env] = "1"
    with fusion.Fuser(gp) as f:
        rs = f.run(sd, decode_threads=nt)
    print("run", i, env, nt, "->", round(rs["frames_total"] / rs["seconds_total"], 1), "frames/s", rs["jpeg_entropy_on_device"], flush=True)
PY
