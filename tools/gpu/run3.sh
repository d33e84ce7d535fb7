#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
cd $GRAFT_REPO_ROOT
python tools/gpu/debug_jpeg.py > gpurun_out/r2c/debug_jpeg.txt 2>&1
cat gpurun_out/r2c/debug_jpeg.txt
( time timeout 1800 python -m pytest tests -m gpu -q -k "not jpeg and not stage_and_cli" ) > gpurun_out/r2c/pytest.log 2>&1
tail -8 gpurun_out/r2c/pytest.log
B="python bench.py --no-cpu-baseline --no-pmc"
( timeout 300 $B --no-single-frame ) > gpurun_out/r2c/b4_batched.json 2> gpurun_out/r2c/b4_batched.err
for v in "pipe_overlap=0" "pipe_overlap=1" "pipe_overlap=1 --tune front_cus=16" "pipe_overlap=1 --tune front_cus=32" "pipe_overlap=1 --tune front_cus=64"; do
  n=$(echo $v | tr ' =-' '___')
  ( timeout 200 $B --steps 1200 --single-frame --tune $v ) > gpurun_out/r2c/sf_$n.json 2> gpurun_out/r2c/sf_$n.err
done
for v in "nt=0" "nt=1" "nt=1 --tune pipe_wgs=2" "nt=0 --tune pipe_overlap=0" "nt=1 --tune pipe_overlap=0"; do
  n=$(echo $v | tr ' =-' '___')
  ( timeout 300 $B --config 1mm --steps 64 --single-frame --tune $v ) > gpurun_out/r2c/mm_$n.json 2> gpurun_out/r2c/mm_$n.err
done
( timeout 300 $B --config 1mm --no-single-frame ) > gpurun_out/r2c/mm_batched.json 2> gpurun_out/r2c/mm_batched.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2c/*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], "us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "ceil", (r.get("pattern_ceiling") or {}).get("rmw_copy_GBs"))
    except Exception as e:
        print(f, "ERR", e)
PY
