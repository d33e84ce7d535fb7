#!/bin/bash
# round 6: k_alloc with the lane-per-block scan, the workgroup-wide heap pop and up to 8 windows per frame: 1 mm and 4 mm rates, 2 / 3 persistent integrate
# workgroups per CU, the queue at 4 096 and 3 072 entries; the whole GPU suite on the shipped build last
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zf
mkdir -p $O
cd $R
bld() { touch scannet_amd/csrc/fuser.hip; SCANFUSE_BUILD_FLAGS="$1" python -c "from scannet_amd import build as b; b.build()" > $O/build.log 2>&1 || tail -5 $O/build.log; }
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("  value", d["value"], "| kernel us", r.get("avg_kernel_us"), "| single_frame", d.get("roofline_single_frame"), "| ooc", r.get("hbm_out_of_cache"))
PY
}
run1mm() { timeout 600 python bench.py --config 1mm --no-cpu-baseline --no-pmc $2 > $O/bench_1mm_$1.json 2> $O/bench_1mm_$1.err; cp bench_detail.json $O/detail_1mm_$1.json; show $O/bench_1mm_$1.json; }
bld ""
echo "== queue 4096, 1 mm"; run1mm q4096 ""
echo "== queue 4096, 1 mm, pipe_wgs=2"; run1mm q4096_pw2 "--tune pipe_wgs=2"
echo "== 4 mm"; timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-e2e > $O/bench_4mm.json 2> $O/bench_4mm.err; cp bench_detail.json $O/detail_4mm.json; show $O/bench_4mm.json
bld "-DSF_ALLOC6_LIST=3072"
echo "== queue 3072, 1 mm"; run1mm q3072 ""
echo "== queue 3072, 1 mm, pipe_wgs=2"; run1mm q3072_pw2 "--tune pipe_wgs=2"
bld ""
( time timeout 1500 python -m pytest tests -m gpu -q -rs -x ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
