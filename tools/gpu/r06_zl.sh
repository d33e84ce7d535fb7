#!/bin/bash
# round 6: the low-priority front stream beside the persistent kernel (front_prio -1, the default): parity, then 1 mm three times and 4 mm once
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zl
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_tsdf.py -q -x -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
for rep in 1 2 3; do
  timeout 600 python bench.py --config 1mm --no-cpu-baseline --no-pmc > $O/b1_$rep.json 2> $O/b.err
  python - $O/b1_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("1 mm: value %.1f single_frame %s" % (d["value"], d.get("roofline_single_frame")))
PY
done
timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-e2e > $O/bench_4mm.json 2> $O/bench_4mm.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06zl/bench_4mm.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("4 mm: value", d["value"], "| kernel us", r.get("avg_kernel_us"), "| single_frame", d.get("roofline_single_frame"), "| ooc", r.get("hbm_out_of_cache"))
PY
