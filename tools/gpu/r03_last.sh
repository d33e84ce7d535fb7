#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence2
mkdir -p $O
cd $R
( timeout 200 python bench.py --config 1mm ) > $O/bench_1mm.json 2> $O/bench_1mm.err; tail -c 300 $O/bench_1mm.err
cd /tmp
