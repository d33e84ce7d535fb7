#!/bin/bash
# the round's judged evidence: everything here is copied into profiles/r02_*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( timeout 900 python bench.py --config 1mm ) > $O/bench_1mm.json 2> $O/bench_1mm.err
( timeout 900 python bench.py --config partition ) > $O/bench_partition.json 2> $O/bench_partition.err
( timeout 900 python bench.py --config scans --steps 12 --host-stage none ) > $O/bench_scans_none.json 2> $O/bench_scans_none.err
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu-decimate ) > $O/bench_scans_gpu_decimate.json 2> $O/bench_scans_gpu_decimate.err
( timeout 900 python bench.py --config scans --steps 12 --host-stage full ) > $O/bench_scans_full.json 2> $O/bench_scans_full.err
cd /tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --teardown > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt.txt 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$n -o pmc -- python $R/bench.py --no-cpu-baseline --no-pmc --no-single-frame --no-colour --steps 320 --teardown > $O/pmc_$n.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm_$n -name "*.db" | head -1) > $O/pmc_$n.txt 2>&1
done
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2final/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; c = j.get("roofline_colour") or {}; s1 = j.get("roofline_single_frame") or {}
        print(f.split("/")[-1], j["value"], j["unit"], "| us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "hbm", r.get("hbm_frac"), "| colour", c.get("frames_per_s"), "| single:", s1.get("frames_per_s"), s1.get("avg_kernel_us"), s1.get("frac"), "| idle", j.get("gpu_idle_pct"), (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json", ".err")).read()[-800:])
PY
head -8 $O/kt.txt | cut -c1-150
