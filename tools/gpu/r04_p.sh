#!/bin/bash
# Round 4: counters of the two inflate kernels alone (tools/gpu/inflate_bench.py), in their own pass -- never together with a trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp
rm -rf /tmp/pmz; timeout 300 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d /tmp/pmz -o pmc -- python $R/tools/gpu/inflate_bench.py --repeats 5 > $O/pmc_inflate.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/pmz -name "*.db" | head -1) > $O/pmc_inflate.txt 2>&1
grep -n "k_inflate" $O/pmc_inflate.txt | cut -c1-200
tail -2 $O/pmc_inflate.log | cut -c1-300
