#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2o
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^E  " $O/pytest.log | head
( timeout 900 python bench.py --no-cpu-baseline ) > $O/bench_4mm.json 2> $O/bench_4mm.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r2o/bench_4mm.json").read().splitlines() if l.startswith("{")][0])
r = j.get("roofline") or {}; c = j.get("roofline_colour") or {}; s1 = j.get("roofline_single_frame") or {}
print(j["value"], "us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "instr/vf", (r.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"), "| colour", c.get("frames_per_s"), c.get("avg_kernel_us"), "| single:", s1.get("frames_per_s"), s1.get("avg_kernel_us"), s1.get("frac"))
PY
tail -3 $O/bench_4mm.err
