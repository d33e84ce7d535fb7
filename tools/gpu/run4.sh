#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
cd $GRAFT_REPO_ROOT
python tools/gpu/debug_jpeg.py > gpurun_out/r2d/debug_jpeg.txt 2>&1
grep -c "differing bytes 0 " gpurun_out/r2d/debug_jpeg.txt
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/r2d/pytest.log 2>&1
tail -6 gpurun_out/r2d/pytest.log
rocprofv3 -L > gpurun_out/r2d/counters_avail.txt 2>&1
B="python bench.py --no-cpu-baseline --no-pmc --no-single-frame --steps 320"
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$n -o pmc -- $B --teardown > $GRAFT_REPO_ROOT/gpurun_out/r2d/pmc_$n.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pm_$n -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r2d/pmc_$n.txt 2>&1
done
cd $GRAFT_REPO_ROOT
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r2d/bench_4mm.json 2> gpurun_out/r2d/bench_4mm.err
tail -c 400 gpurun_out/r2d/bench_4mm.json
