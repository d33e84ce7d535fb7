#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run" ) 2>&1 | tail -2
timeout 300 python tools/gpu/inflate_bench.py 2>&1 | tee gpurun_out/r04n/inflate_bench.json
( timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --threads 4 --out gpurun_out/r04n/e2e.json ) > gpurun_out/r04n/e2e.log 2>&1
python -c "
import json; print(json.load(open('gpurun_out/r04n/e2e.json'))['fuse'])"
