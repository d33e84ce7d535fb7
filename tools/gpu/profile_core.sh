#!/bin/bash
# kernel trace + counter passes of the default stream (the profile DESIGN.md quotes per kernel); ~3 GPU-minutes
#   gpurun --timeout 1200 -- 'bash tools/gpu/profile_core.sh <outdir-under-gpurun_out>'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
Q="--no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --repeats 1 --teardown"
cd /tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py $Q > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt.txt 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$n -o pmc -- python $R/bench.py $Q --no-single-frame --no-colour --steps 320 > $O/pmc_$n.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm_$n -name "*.db" | head -1) > $O/pmc_$n.txt 2>&1
done
head -16 $O/kt.txt | cut -c1-170
grep -E "k_alloc|k_integrate<1, false" $O/pmc_SQ_WAVES.txt | cut -c1-200
