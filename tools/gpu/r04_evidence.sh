#!/bin/bash
# The measurements DESIGN.md section 5 quotes for round 4, in one gpurun call (~20 GPU-minutes); the files are copied into profiles/r04_* by hand.
# The profiler runs come last: a profiler that takes a process down must not take the other measurements with it.
#   gpurun --timeout 3000 -- 'bash tools/gpu/r04_evidence.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ev
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
( timeout 900 python bench.py --config 1mm ) > $O/bench_1mm.json 2> $O/bench_1mm.err
( timeout 900 python bench.py --config partition ) > $O/bench_partition.json 2> $O/bench_partition.err
( timeout 900 python bench.py --config scans --steps 12 ) > $O/bench_scans_gpu.json 2> $O/bench_scans_gpu.err
( timeout 600 python bench.py --gpus 2 --share-gpu --steps 64 --warmup 5 --no-pmc ) > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err
( timeout 900 python tools/e2e_bench.py --frames 5578 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
cd /tmp
# the driver's command under the kernel trace, then the default command (timeline of the passes)
rm -rf /tmp/ktd; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktd -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-e2e --no-single-frame --no-out-of-cache --no-depth-only --teardown > $O/kt_driver.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/ktd -name "*.db" | head -1) > $O/kt_driver.txt 2>&1
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-single-frame --no-out-of-cache --no-depth-only --repeats 1 --teardown > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt.txt 2>&1
python $R/tools/timeline.py $(find /tmp/kt -name "*.db" | head -1) > $O/timeline.txt 2>&1
# counters in their own passes (never together with a trace): the RGB-D pass
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$n -o pmc -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-depth-only --repeats 1 --steps 320 --teardown > $O/pmc_$n.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm_$n -name "*.db" | head -1) > $O/pmc_$n.txt 2>&1
done
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04ev/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; d = j.get("roofline_depth_only") or {}; s1 = j.get("roofline_single_frame") or {}
        print(f.split("/")[-1], j["value"], j["unit"], "depth-only", j.get("value_depth_only"), "| us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "hbm", r.get("hbm_frac"),
              "| depth us", d.get("avg_kernel_us"), d.get("frac"), "| single:", s1.get("frames_per_s"), s1.get("avg_kernel_us"), s1.get("frac"), "| ooc", (r.get("hbm_out_of_cache") or {}).get("frac"),
              "| e2e", (j.get("end_to_end") or {}).get("frames_per_s"), (j.get("end_to_end_rgbd") or {}).get("frames_per_s"), "| parity", (j.get("parity") or {}).get("sha256_equal"),
              "| idle", j.get("gpu_idle_pct"), (j.get("cpu_baseline") or {}).get("value"), "| prefix", (j.get("prefix_check") or {}).get("sha256_equal"), j.get("per_rank_frames_per_s"))
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json", ".err")).read()[-600:])
for f in sorted(glob.glob("gpurun_out/r04ev/e2e*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], j["fuse"]["frames_per_s_end_to_end"], {k: j[k] for k in j if k.endswith("_s")}, j.get("marching_cubes_phases_ms", {}).get("total"))
PY
head -12 $O/kt_driver.txt | cut -c1-160
