#!/bin/bash
# Round 6, the evidence behind DESIGN.md section 5 in one call (MI355X, ~15 GPU-minutes); the files are copied into profiles/r06_* by hand.
# The profiler runs come last: a profiler that takes a process down must not take the other measurements with it.
#   gpurun --timeout 3000 -- 'bash tools/gpu/r06_final.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -rs ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time SF_BENCH_DETAIL=$O/detail_4mm.json timeout 1200 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( time SF_BENCH_DETAIL=$O/detail_4mm_driver_args.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
( timeout 900 python bench.py --config 1mm ) > $O/bench_1mm.json 2> $O/bench_1mm.err
( timeout 900 python bench.py --config scans --steps 12 --host-stage gpu ) > $O/bench_scans_gpu.json 2> $O/bench_scans_gpu.err
( timeout 900 python bench.py --config partition ) > $O/bench_partition.json 2> $O/bench_partition.err
( timeout 600 python bench.py --gpus 2 --share-gpu --steps 64 --warmup 5 --repeats 3 --no-pmc ) > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err
cd /tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --repeats 1 --teardown > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt.txt 2>&1
python $R/tools/timeline.py $(find /tmp/kt -name "*.db" | head -1) --skip k_synth --skip at:: > $O/timeline.txt 2>&1
rm -rf /tmp/kt2; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --teardown > $O/kt_driver.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt2 -name "*.db" | head -1) > $O/kt_driver.txt 2>&1
rm -rf /tmp/kt3; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o kt -- python $R/tools/e2e_bench.py --frames 2048 --color jpeg --color-res 1296x968 --fuse-only --threads 4 > $O/kt_e2e_rgbd.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt3 -name "*.db" | head -1) > $O/kt_e2e_rgbd.txt 2>&1
python $R/tools/timeline.py $(find /tmp/kt3 -name "*.db" | head -1) --skip k_synth --skip at:: > $O/timeline_e2e_rgbd.txt 2>&1
cd $R
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06final/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; e = j.get("end_to_end") or {}; g = j.get("end_to_end_rgbd") or {}; cal = r.get("valu_peak_calibration") or {}
        print(f.split("/")[-1], j["value"], j["unit"], j.get("value_depth_only"), "| us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), cal.get("frac_overlap_floor"), r.get("issue_ratio_4_cycles"), "hbm", r.get("hbm_frac"),
              "| ooc", (r.get("hbm_out_of_cache") or {}).get("frac"), "| e2e", e.get("frames_per_s"), e.get("frames_per_s_first_and_second_run"), "| rgbd", g.get("frames_per_s"), g.get("frames_per_s_first_and_second_run"),
              "| single", (j.get("roofline_single_frame") or {}).get("frames_per_s"), "| cpu", (j.get("cpu_baseline") or {}).get("value"), "| parity", (j.get("parity") or {}).get("sha256_equal"))
    except Exception as ex:
        print(f, "ERR", ex); print(open(f.replace(".json", ".err")).read()[-600:])
PY
head -9 $O/kt.txt | cut -c1-150; head -9 $O/kt_driver.txt | cut -c1-150
