#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
( timeout 600 python bench.py --no-cpu-baseline ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( timeout 600 python bench.py --no-cpu-baseline --config 1mm ) > $O/bench_1mm.json 2> $O/bench_1mm.err
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$n -o pmc -- python $R/bench.py --no-cpu-baseline --no-pmc --no-single-frame --steps 320 --teardown > $O/pmc_$n.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm_$n -name "*.db" | head -1) > $O/pmc_$n.txt 2>&1
done
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --teardown > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt.txt 2>&1
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2e/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; s1 = j.get("roofline_single_frame") or {}
        print(f.split("/")[-1], j["value"], "us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "instr/vf", (r.get("valu_detail") or {}).get("valu_insts_per_voxel_frame"), "hbm", r.get("hbm_frac"),
              "| single:", s1.get("frames_per_s"), s1.get("avg_kernel_us"), s1.get("frac"), (s1.get("pattern_ceiling") or {}).get("rmw_copy_GBs"))
    except Exception as e:
        print(f, "ERR", e)
PY
