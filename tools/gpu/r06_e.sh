#!/bin/bash
# round 6, fifth call: the allocation walk without nested branches, sf_fuse_run_prepare (first run of an RGB-D scan), evidence for profiles/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06e
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_pipeline.py tests/test_zz_depthsensing_ranks.py -m gpu -q -x -rsxX ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
Q="--no-cpu-baseline --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --repeats 3"
( SF_BENCH_DETAIL=$O/detail.json timeout 600 python bench.py $Q ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.load(open("$O/detail.json")); r=j["roofline"]
print("default", j["value"], "kernel us", r["avg_kernel_us"], "front", {k:(v.get("avg_us_alone"), v.get("insts_valu")) for k,v in (r.get("front_chain") or {}).items()})
PY
for k in 0 1; do
  ( HIP_FORCE_DEV_KERNARG=$k SF_BENCH_DETAIL=$O/detail_devkernarg$k.json timeout 300 python bench.py $Q --no-pmc ) > $O/bench_devkernarg$k.json 2> $O/bench_devkernarg$k.err
  python -c "
import json; j=json.load(open('$O/detail_devkernarg$k.json')); print('HIP_FORCE_DEV_KERNARG=$k', j['value'], j['repeats']['value_min'], j['repeats']['value_max'], 'kernel us', j['roofline']['avg_kernel_us'])"
done
for extra in "" "--no-prepare"; do
  ( SF_RUN_TIMING=1 timeout 600 python tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 $extra ) > "$O/e2e_rgbd$extra.log" 2>&1
  echo "e2e rgbd $extra"; grep "sf_fuse_run:" "$O/e2e_rgbd$extra.log" | cut -c1-330; tail -1 "$O/e2e_rgbd$extra.log" | cut -c250-700
done
( time SF_BENCH_DETAIL=$O/detail_driver.json timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 3800 $O/bench_driver.json; tail -3 $O/bench_driver.err
cd /tmp
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --teardown > $O/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/rocprofv3_kernel_stats.txt 2>&1; head -30 $O/rocprofv3_kernel_stats.txt | cut -c1-180
