#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "gpu_inflate" ) 2>&1 | tail -8
