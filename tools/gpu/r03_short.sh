#!/bin/bash
# why is a 20-step run from an empty volume slow?  A/B over scene / noise and a kernel trace of the short run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
Q="--no-pmc --no-e2e --no-cpu-baseline --no-out-of-cache --no-single-frame --no-colour --steps 20 --warmup 5 --repeats 200"
( timeout 600 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_tsdf.log; cat $O/pytest_tsdf.log
for sn in "1 2" "0 1" "0 2" "1 1"; do
  set -- $sn
  timeout 200 python bench.py $Q --scene $1 --noise $2 > $O/short_s$1_n$2.json 2> $O/short_s$1_n$2.err
done
timeout 200 python bench.py $Q --tune ramp=0 > $O/short_ramp0.json 2> $O/short_ramp0.err
cd /tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py $Q --repeats 20 --teardown > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt_short.txt 2>&1
cd $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/short*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0]); r=j["roofline"]
        print(f.split("/")[-1], j["value"], j["repeats"]["value_min"], j["repeats"]["value_max"], "kernel us", r["avg_kernel_us"], r["launches"], j["config"]["blocks_live_end"])
    except Exception as e: print(f, "ERR", e)
PY
head -14 $O/kt_short.txt | cut -c1-160
cd $R
for ar in 1 0; do
  timeout 300 python bench.py --no-pmc --no-e2e --no-cpu-baseline --no-out-of-cache --no-colour --repeats 3 --tune alloc_ray=$ar > $O/full_ar$ar.json 2> $O/full_ar$ar.err
  timeout 200 python bench.py $Q --tune alloc_ray=$ar > $O/short_ar$ar.json 2> $O/short_ar$ar.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/*_ar*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0]); r=j["roofline"]; s1=j.get("roofline_single_frame") or {}
        print(f.split("/")[-1], j["value"], "kernel us", r["avg_kernel_us"], "single", s1.get("frames_per_s"), s1.get("avg_kernel_us"))
    except Exception as e: print(f, "ERR", e)
PY
