#!/bin/bash
# round 6: the presence cache at 1 mm, one frame per launch: probe counts and the kernels of a frame with the cache on and off
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06za
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_tsdf.py -q -x -m gpu -k "presence_cache and cube64" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for bc in 1 0; do
  SF_PROBE_BRICK_CACHE=$bc timeout 300 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -4
  cd /tmp; rm -rf /tmp/kt; SF_PROBE_BRICK_CACHE=$bc SF_PROBE_ONLY_BATCH1=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p$bc.log 2>&1
  echo "== brick_cache $bc"; python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep -E "k_alloc|k_compactify|k_integrate|k_prepass" | cut -c1-160; cd $R
done 2>&1 | tee $O/kernels.txt
