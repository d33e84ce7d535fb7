#!/bin/bash
# round 2, GPU run 1: parity suite, then the bench configurations
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a/pytest.log 2>&1
tail -5 gpurun_out/r2a/pytest.log
( time timeout 600 python bench.py ) > gpurun_out/r2a/bench_4mm.json 2> gpurun_out/r2a/bench_4mm.err
tail -c 600 gpurun_out/r2a/bench_4mm.json
( time timeout 420 python bench.py --config 1mm ) > gpurun_out/r2a/bench_1mm.json 2> gpurun_out/r2a/bench_1mm.err
tail -c 600 gpurun_out/r2a/bench_1mm.json
( time timeout 200 python bench.py --steps 1200 --single-frame --no-pmc --no-cpu-baseline --tune pipe_overlap=0 ) > gpurun_out/r2a/bench_sf_nooverlap.json 2> gpurun_out/r2a/bench_sf_nooverlap.err
( time timeout 200 python bench.py --steps 1200 --single-frame --no-pmc --no-cpu-baseline ) > gpurun_out/r2a/bench_sf_overlap.json 2> gpurun_out/r2a/bench_sf_overlap.err
( time timeout 300 python bench.py --config partition --scan-frames 20000 --stripes-at-one ) > gpurun_out/r2a/bench_partition.json 2> gpurun_out/r2a/bench_partition.err
( time timeout 400 python bench.py --config scans --steps 4 --host-stage clean ) > gpurun_out/r2a/bench_scans.json 2> gpurun_out/r2a/bench_scans.err
tail -c 300 gpurun_out/r2a/bench_sf_nooverlap.json gpurun_out/r2a/bench_sf_overlap.json gpurun_out/r2a/bench_partition.json gpurun_out/r2a/bench_scans.json
