#!/bin/bash
# Round 5, call J: the f1 identity tests (GPU decimation == its sequential restatement), the new multi-GPU tests (skip on one GPU), the colour pipeline tests.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_simplify_gpu.py -q -x -k "restatement" -s ) > $O/pytest_f1.log 2>&1; tail -15 $O/pytest_f1.log
( time timeout 600 python -m pytest tests/test_multi_gpu.py -q -rs ) > $O/pytest_multi.log 2>&1; tail -8 $O/pytest_multi.log
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "colour" ) > $O/pytest_colour.log 2>&1; tail -5 $O/pytest_colour.log
