#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run" ) 2>&1 | tail -3
timeout 300 python tools/gpu/inflate_bench.py 2>&1 | tail -1
