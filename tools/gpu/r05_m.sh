#!/bin/bash
# Round 5, call M: the phase clock of a depth-only scan's first and second run in a process (SF_RUN_TIMING)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m
mkdir -p $O
cd /tmp
( SF_RUN_TIMING=1 timeout 600 python $R/tools/e2e_bench.py --frames 5578 --fuse-only --threads 4 --out $O/e2e.json ) > $O/e2e.log 2>&1; grep -E "sf_fuse_run|frames_per_s" $O/e2e.log | cut -c1-460
