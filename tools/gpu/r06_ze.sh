#!/bin/bash
# round 6: k_alloc's drain with one heap pop per workgroup: parity of the allocation tests, then the workgroup timing again (the shipped .so is the -DSF_ALLOC_TIMING build)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06ze
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tsdf.py -q -x -m gpu -k "presence_cache or allocation_kernels or one_mm or furnished or garbage" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for t in "overlap=0" ""; do echo "== timing build, tune: $t"; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -28; done
