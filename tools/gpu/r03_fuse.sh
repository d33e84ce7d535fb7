#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
( timeout 400 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -k "clamp_match or one_frame or four_schedules or plane_bit or small_weight or invalid_pose or ragged" 2>&1 | grep -E "passed|failed|rror" | tail -3 )
Q="--no-pmc --no-e2e --no-cpu-baseline --no-out-of-cache --no-colour --repeats 1 --steps 464 --warmup 64"
for t in "" "--tune prepass_fuse=0"; do
  timeout 200 python bench.py $Q $t > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/x.json") if l.startswith("{")][0]); s=j["roofline_single_frame"]; print("[$t] single", s["frames_per_s"], s["avg_kernel_us"], s["frac"], "host", s["live_stream_host_buffers"]["frames_per_s"])
except Exception as e: print("[$t] ERR", e, open("$O/x.err").read()[-300:])
PY
done
