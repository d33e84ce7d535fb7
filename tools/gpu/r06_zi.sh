#!/bin/bash
# round 6: k_compactify_few (one list-counter atomic per 2 048 entries): parity, then the 1 mm one-frame-per-launch timeline and the rates
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zi
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py -q -x -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
cd /tmp
for t in "" "pipe_wgs=1" "pipe_wgs=2"; do
  rm -rf /tmp/kt; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p.log 2>&1
  echo "== tune: $t | $(grep -o 'fps [0-9.]*' $O/p.log | tail -1)"; python $R/tools/gpu/period_summary.py $(find /tmp/kt -name "*.db" | head -1)
done 2>&1 | tee $O/matrix.txt
cd $R
timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-e2e > $O/bench_4mm.json 2> $O/bench_4mm.err; cp bench_detail.json $O/detail_4mm.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06zi/bench_4mm.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("4 mm: value", d["value"], "| kernel us", r.get("avg_kernel_us"), "| single_frame", d.get("roofline_single_frame"), "| ooc", r.get("hbm_out_of_cache"))
PY
