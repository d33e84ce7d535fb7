#!/bin/bash
# round 6: the presence cache of the allocation kernels -- parity first, then the 1 mm and 4 mm rates with the cache on and off
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06z
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tsdf.py -q -x -m gpu -k "presence_cache or allocation_kernels or garbage or one_mm or gc_ragged or reset_gives or furnished" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for bc in 1 0; do
  echo "== 1 mm, brick_cache=$bc"
  timeout 600 python bench.py --config 1mm --no-cpu-baseline --no-pmc --tune brick_cache=$bc > $O/bench_1mm_bc$bc.json 2> $O/bench_1mm_bc$bc.err; cp bench_detail.json $O/detail_1mm_bc$bc.json
  python - <<PY
import json
d=json.loads(open("$O/bench_1mm_bc$bc.json").read().strip().splitlines()[-1])
print("value", d["value"], "single_frame", d.get("roofline_single_frame"), "ooc", d["roofline"].get("hbm_out_of_cache"))
PY
done
for bc in 1 0; do
  echo "== 4 mm, brick_cache=$bc"
  timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-e2e --tune brick_cache=$bc > $O/bench_4mm_bc$bc.json 2> $O/bench_4mm_bc$bc.err; cp bench_detail.json $O/detail_4mm_bc$bc.json
  python - <<PY
import json
d=json.loads(open("$O/bench_4mm_bc$bc.json").read().strip().splitlines()[-1])
print("value", d["value"], "single_frame", d.get("roofline_single_frame"), "front", d["roofline"].get("front_chain_us_alone"))
PY
done
