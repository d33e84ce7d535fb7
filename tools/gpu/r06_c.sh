#!/bin/bash
# round 6, third call: the device-to-device exchange of bin/depthsensing --ranks, the compaction with its frame constants from LDS, the decimation tolerance
# test, the drop-in test with the reference Segmentator; the end-to-end RGB-D timeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_zz_depthsensing_ranks.py tests/test_simplify_gpu.py tests/test_gpu_tsdf.py "tests/test_gpu_pipeline.py::test_drop_in_executables" "tests/test_gpu_pipeline.py::test_bench_with_two_ranks_sharing_one_gpu" -m gpu -q -rsxX -s ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head; grep -E "faces_ratio" $O/pytest.log | cut -c1-600
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --repeats 3"
( SF_BENCH_DETAIL=$O/detail.json timeout 600 python bench.py $Q ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.load(open("$O/detail.json")); r=j["roofline"]
print(j["value"], "kernel us", r["avg_kernel_us"], "front", {k:(v.get("avg_us_alone"), v.get("insts_valu")) for k,v in (r.get("front_chain") or {}).items()})
PY
( SF_BENCH_DETAIL=$O/detail_driver.json timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --no-pmc ) > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-200 $O/bench_driver.json
# end-to-end RGB-D: timeline of the steady state
cd /tmp
rm -rf /tmp/kt_e2e; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_e2e -o kt -- python $R/tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 > $O/e2e_kt.log 2>&1
tail -2 $O/e2e_kt.log | cut -c1-400
DB=$(find /tmp/kt_e2e -name "*.db" | head -1)
python $R/tools/timeline.py $DB -700 260 > $O/timeline_e2e_rgbd.txt 2>&1
python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_e2e_rgbd.txt 2>&1; head -25 $O/kernel_stats_e2e_rgbd.txt | cut -c1-200
