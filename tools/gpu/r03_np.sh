#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --repeats 5"
for v in "" "--no-profile" "--no-profile --tune xcd_walk=0" "--tune overlap=0"; do
  timeout 200 python bench.py $Q $v > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/x.json") if l.startswith("{")][0]); print("[$v]", j["value"], j["repeats"]["value_min"], j["repeats"]["value_max"], (j["roofline"] or {}).get("avg_kernel_us"))
except Exception as e: print("[$v]","ERR",e, open("$O/x.err").read()[-300:])
PY
done
