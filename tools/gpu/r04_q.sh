#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04q/bench_4mm.json").read().splitlines() if l.startswith("{")][0])
e = j["end_to_end"]
print(j["value"], j.get("value_depth_only"), j["roofline"]["frac"], j["roofline"]["avg_kernel_us"], e["frames_per_s"], e["frames_per_s_first_and_second_run"], {k: v for k, v in e["inflate_kernels"].items() if k != "what"}, e["host_inflate"], (j.get("end_to_end_rgbd") or {}).get("frames_per_s"), j["parity"]["sha256_equal"])
PY
tail -3 $O/bench_4mm.err
