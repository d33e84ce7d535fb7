#!/bin/bash
# round 4, fourth call: pass ramp A/B on the driver's command, marching-cubes download with transparent huge pages
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
X="--no-pmc --no-cpu-baseline --no-single-frame --no-out-of-cache --no-e2e --no-profile"
for t in "ramp=8" "ramp=4" "ramp=2" "ramp=4 --tune ramp_geo=0" "ramp=6"; do
  n=$(echo "$t" | tr -d ' -' | tr '=' '_')
  ( timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $X --tune $t ) > $O/ramp_$n.json 2> $O/ramp_$n.err
  ( timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $X --depth-only --tune $t ) > $O/rampd_$n.json 2> $O/rampd_$n.err
done
( timeout 300 python bench.py $X --no-depth-only --tune ramp=4 ) > $O/full_ramp_4.json 2> $O/full_ramp_4.err
( timeout 300 python bench.py $X --no-depth-only --tune ramp=8 ) > $O/full_ramp_8.json 2> $O/full_ramp_8.err
( timeout 600 python tools/e2e_bench.py --frames 5578 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04d/*ramp*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        print(f.split("/")[-1], j["value"], j["repeats"]["value_min"], j["repeats"]["value_max"])
    except Exception as e:
        print(f, "ERR", e)
j = json.load(open("gpurun_out/r04d/e2e_5578_gpu.json"))
print("e2e", j["fuse"], "mc", j["marching_cubes_s"], j["marching_cubes_second_call_s"], j["marching_cubes_phases_ms"])
PY
