#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
mkdir -p $O
cd $R
for ramp in 0 2 4 8; do
  for rep in 1 2 3; do
    ( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-single-frame --no-colour --tune ramp=$ramp ) > $O/short_ramp${ramp}_$rep.json 2> $O/short.err
  done
done
( timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-single-frame --no-colour --tune ramp=4 ) > $O/full_ramp4.json 2>> $O/short.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2p/*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        print(f.split("/")[-1], j["value"], "ms/step", j["ms_per_step"], "enq", j["host_enqueue_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -2 $O/short.err
