#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "inflate or fuse_run or drop_in or shard" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for t in 16 4; do
( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --threads $t --out $O/e2e_t$t.json ) > $O/e2e_t$t.log 2>&1
grep "sf_fuse_run" $O/e2e_t$t.log | tail -2 | cut -c1-220
python -c "
import json; print($t, json.load(open('gpurun_out/r04j/e2e_t$t.json'))['fuse']['frames_per_s_end_to_end'])"
done
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
python - <<'PY'
import json
for n in ("e2e_colour_raw", "e2e_colour_jpeg_1296"):
    print(n, json.load(open("gpurun_out/r04j/%s.json" % n))["fuse"])
PY
