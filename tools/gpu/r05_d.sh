#!/bin/bash
# (run while the packed pairs were the default build: -DSF_SCALAR_PAIRS selected the plain pairs then; since then plain pairs are the default and -DSF_PACKED_PAIRS selects the round-4 kernels)
# Round 5, call D: counters of the RGB-D integrate kernel with the texel gather, packed against plain pairs; the list of counters the box offers.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
ARGS="--no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --repeats 1 --steps 320 --teardown"
run_sets() {
  tag=$1
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
             "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
    i=$((i+1))
    rm -rf /tmp/pm_$tag$i
    timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$tag$i -o pmc -- python $R/bench.py $ARGS > $O/pmc_${tag}_$i.log 2>&1
    python $R/tools/rocpd_summary.py $(find /tmp/pm_$tag$i -name "*.db" | head -1) 2>&1 | grep -E "k_integrate<1, 2|k_alloc_ray|k_prepass" > $O/pmc_${tag}_$i.txt
  done
  cat $O/pmc_${tag}_*.txt | cut -c1-40,72-200
}
cd $R
run_sets packed
export SCANFUSE_BUILD_FLAGS="-DSF_SCALAR_PAIRS -fno-slp-vectorize"
python -c "from scannet_amd import build; build.build(force=True)" > $O/build_scalar.log 2>&1
run_sets scalar
