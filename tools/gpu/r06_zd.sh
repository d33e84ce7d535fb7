#!/bin/bash
# round 6: where a workgroup of k_alloc<6> spends its time (-DSF_ALLOC_TIMING), beside the integrate kernel and alone; then the queue at 3 072 entries (61 KiB of LDS:
# fits beside TWO persistent integrate workgroups of 48 KiB) with 3 and 2 persistent workgroups per CU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zd
mkdir -p $O
cd $R
bld() { touch scannet_amd/csrc/fuser.hip; SCANFUSE_BUILD_FLAGS="$1" python -c "from scannet_amd import build as b; b.build()" > $O/build.log 2>&1 || tail -5 $O/build.log; }
bld "-DSF_ALLOC_TIMING"
for t in "" "overlap=0"; do echo "== timing build, tune: $t"; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -2; done
bld "-DSF_ALLOC6_LIST=3072"
for t in "" "pipe_wgs=2" "pipe_wgs=2,front_cus=32" ; do echo "== queue 3072, tune: $t"; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -1; done
bld ""
for t in "" "pipe_wgs=2" "pipe_wgs=1"; do echo "== queue 4096, tune: $t"; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 python tools/gpu/alloc_1mm_probe.py 2>&1 | tail -1; done
