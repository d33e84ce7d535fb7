#!/bin/bash
# A/B of scheduling switches on the full stream: bash tools/gpu/r03_tune.sh <outdir> "k=v[,k2=v2]" ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
for t in "$@"; do
  args=""; for kv in ${t//,/ }; do args="$args --tune $kv"; done
  timeout 200 python bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --repeats 3 $args > $O/tune_$t.json 2> $O/tune_$t.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/tune_$t.json") if l.startswith("{")][0]); print("$t", j["value"], j["repeats"]["value_min"], j["repeats"]["value_max"], "integrate us", j["roofline"]["avg_kernel_us"])
except Exception as e: print("$t","ERR",e, open("$O/tune_$t.err").read()[-300:])
PY
done
