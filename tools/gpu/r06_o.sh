#!/bin/bash
# round 6: the default command and the driver's command after the planes-only JPEG path
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06o
mkdir -p $O
cd $R
( time SF_BENCH_DETAIL=$O/detail_driver.json timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 3900 $O/bench_driver.json; tail -3 $O/bench_driver.err
