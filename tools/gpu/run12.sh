#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2l
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|Error" $O/pytest.log | head
( timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( timeout 900 python bench.py --tune cull=0 --no-pmc --no-cpu-baseline --no-colour ) > $O/bench_4mm_nocull.json 2> $O/bench_4mm_nocull.err
( timeout 900 python bench.py --config 1mm --no-cpu-baseline ) > $O/bench_1mm.json 2> $O/bench_1mm.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2l/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; c = j.get("roofline_colour") or {}; s1 = j.get("roofline_single_frame") or {}
        print(f.split("/")[-1], j["value"], "us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "instr/vf", (r.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"), "hbm", r.get("hbm_frac"), j.get("roofline_inputs"),
              "| colour", c.get("frames_per_s"), c.get("avg_kernel_us"), "| single:", s1.get("frames_per_s"), s1.get("avg_kernel_us"), s1.get("frac"), (s1.get("pattern_ceiling") or {}).get("rmw_copy_GBs"))
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
