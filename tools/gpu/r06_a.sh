#!/bin/bash
# round 6, first call: the GPU suite, smoke, the driver's bench command (compact line), the default command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06a
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR|XPASS|XFAIL" $O/pytest.log | head
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time SF_BENCH_DETAIL=$O/detail_driver.json timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 4200 $O/bench_driver.json; tail -3 $O/bench_driver.err
( time SF_BENCH_DETAIL=$O/detail_default.json timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 4200 $O/bench_default.json; tail -3 $O/bench_default.err
