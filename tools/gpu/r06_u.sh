#!/bin/bash
# round 6: what does k_jpeg_huff wait for?  SQ counters of one 1296x968 picture per launch (separate --pmc passes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06u
mkdir -p $O
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o pm -- python $R/tools/gpu/huff_pmc.py > $O/run_$tag.log 2>&1
  DB=$(find /tmp/pm -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/pmc_kernel.py $DB k_jpeg_huff; else echo "no db for $set"; tail -3 $O/run_$tag.log; fi
done 2>&1 | tee $O/huff_pmc.txt
