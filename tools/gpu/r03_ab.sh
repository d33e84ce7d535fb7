#!/bin/bash
# parity of the fusion kernels + the two headline rates (driver arguments, full stream) + one-frame mode; ~4 GPU-minutes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-ab}
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_tsdf.log; cat $O/pytest_tsdf.log
Q="--no-pmc --no-e2e --no-cpu-baseline --no-out-of-cache --no-colour"
timeout 300 python bench.py $Q --repeats 5 $2 > $O/full.json 2> $O/full.err
timeout 200 python bench.py $Q --no-single-frame --steps 20 --warmup 5 --repeats 300 $2 > $O/short.json 2> $O/short.err
python - <<PY
import json
for f in ("full","short"):
    try:
        j=json.loads([l for l in open("$O/%s.json"%f) if l.startswith("{")][0]); r=j["roofline"]; s1=j.get("roofline_single_frame") or {}
        print(f, j["value"], j["repeats"]["value_min"], j["repeats"]["value_max"], "integrate us", r["avg_kernel_us"], "single", s1.get("frames_per_s"), s1.get("avg_kernel_us"), (s1.get("live_stream_host_buffers") or {}).get("frames_per_s"))
    except Exception as e: print(f, "ERR", e, open("$O/%s.err"%f).read()[-500:])
PY
