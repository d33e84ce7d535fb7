#!/bin/bash
# round 6: who runs beside whom at 1 mm, one frame per launch: the kernel timeline of the last frames of tools/gpu/alloc_1mm_probe.py, 3 and 2 persistent workgroups per CU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zg
mkdir -p $O
cd /tmp
for t in "" "pipe_wgs=2" "pipe_wgs=1"; do
  rm -rf /tmp/kt; SF_PROBE_ONLY_BATCH1=1 SF_PROBE_TUNE=$t timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/gpu/alloc_1mm_probe.py > $O/p.log 2>&1
  echo "== tune: $t"; tail -1 $O/p.log | cut -c1-120
  python $R/tools/timeline.py $(find /tmp/kt -name "*.db" | head -1) -24 24 --skip k_synth --skip at:: | cut -c1-100
done 2>&1 | tee $O/timeline.txt
