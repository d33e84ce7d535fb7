#!/bin/bash
# round 6: what does k_alloc<6> (1 mm voxels) spend 2 ms per frame on?  SQ counters, separate --pmc passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06v
mkdir -p $O
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm; SF_PROBE_ONLY_BATCH1=1 timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu/alloc_1mm_probe.py > $O/run_$tag.log 2>&1
  DB=$(find /tmp/pm -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/pmc_kernel.py $DB "k_alloc<6, false"; else echo "no db for $set"; tail -3 $O/run_$tag.log; fi
done 2>&1 | tee $O/alloc6_pmc.txt
