#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log
( timeout 600 python tools/e2e_bench.py --frames 5578 --out $O/e2e_5578.json ) > $O/e2e_5578.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color jpeg --fuse-only --out $O/e2e_colour_jpeg.json ) > $O/e2e_colour_jpeg.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
tail -n 2 $O/e2e_*.log | cut -c1-900
( timeout 900 python bench.py --config scans --steps 12 --host-stage full ) > $O/bench_scans_full.json 2> $O/bench_scans_full.err
( timeout 400 python bench.py --config scans --steps 12 --host-stage none ) > $O/bench_scans_none.json 2> $O/bench_scans_none.err
tail -c 700 $O/bench_scans_full.json; tail -c 500 $O/bench_scans_none.json; tail -2 $O/bench_scans_full.err
