#!/bin/bash
# round 6: the full GPU suite on the tree with the presence cache, then k_alloc<6> at 1 mm taken apart again with the cache on (measurement builds: the volume is wrong with any switch)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zc
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -rs -x ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
cp scannet_amd/libscanfuse.so /tmp/libscanfuse.keep
for flag in "-DSF_ABLATE_ALLOC_PHASE2" "-DSF_ABLATE_ALLOC_SCAN" "-DSF_ABLATE_ALLOC_WALK"; do
  touch scannet_amd/csrc/fuser.hip
  SCANFUSE_BUILD_FLAGS="$flag" python -c "from scannet_amd import build as b; b.build()" > $O/build.log 2>&1 || tail -5 $O/build.log
  cd /tmp; rm -rf /tmp/kt; SF_PROBE_ONLY_BATCH1=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p.log 2>&1
  echo "== flags: $flag"; python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep -E "k_alloc<6|k_compactify|k_integrate_pipe" | cut -c1-140; cd $R
done 2>&1 | tee $O/ablate.txt
