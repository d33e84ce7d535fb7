#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/gpu/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost > gpurun_out/r2b/launch_cost.json 2>&1
cat gpurun_out/r2b/launch_cost.json
python tools/gpu/debug_jpeg.py > gpurun_out/r2b/debug_jpeg.txt 2>&1
cat gpurun_out/r2b/debug_jpeg.txt
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/r2b/pytest.log 2>&1
tail -15 gpurun_out/r2b/pytest.log
( time timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r2b/bench_4mm.json 2> gpurun_out/r2b/bench_4mm.err
tail -c 300 gpurun_out/r2b/bench_4mm.json
( time timeout 400 python bench.py --config scans --steps 4 --host-stage clean ) > gpurun_out/r2b/bench_scans.json 2> gpurun_out/r2b/bench_scans.err
tail -c 600 gpurun_out/r2b/bench_scans.json; tail -3 gpurun_out/r2b/bench_scans.err
