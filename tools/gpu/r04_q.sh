#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04q/bench_4mm_driver_args.json").read().splitlines() if l.startswith("{")][0])
e = j["end_to_end"]
print(j["value"], j.get("value_depth_only"), e["frames_per_s"], e["frames_per_s_first_and_second_run"], e["inflate_kernels"], e["host_inflate"], (j.get("end_to_end_rgbd") or {}).get("frames_per_s"), j["parity"]["sha256_equal"])
PY
tail -3 $O/bench_4mm_driver_args.err
