#!/bin/bash
# round 6: k_alloc<6> with the window centred on the box around the tile's ray segments
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zb
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tsdf.py -q -x -m gpu -k "(presence_cache and cube64) or one_mm" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
SF_PROBE_BRICK_CACHE=1 timeout 300 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -3
cd /tmp; rm -rf /tmp/kt; SF_PROBE_ONLY_BATCH1=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep -E "k_alloc|k_compactify|k_integrate|k_prepass" | cut -c1-160; cd $R
