#!/bin/bash
# round 6: 1 mm, one frame per launch: the front stream on CUs of its own (front_cus) x persistent integrate workgroups per CU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zh
mkdir -p $O
cd /tmp
for t in "" "pipe_wgs=1" "front_cus=16" "front_cus=32" "front_cus=48" "front_cus=64" "front_cus=32,pipe_wgs=2" "front_cus=64,pipe_wgs=2" "front_cus=24" "front_cus=40"; do
  rm -rf /tmp/kt; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p.log 2>&1
  echo "== tune: $t | $(grep -o 'fps [0-9.]*' $O/p.log | tail -1)"; python $R/tools/gpu/period_summary.py $(find /tmp/kt -name "*.db" | head -1)
done 2>&1 | tee $O/matrix.txt
