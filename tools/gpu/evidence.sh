#!/bin/bash
# The measurements DESIGN.md section 5 quotes, in one gpurun call (MI355X, ~14 GPU-minutes); the files are copied into profiles/rNN_* by hand.
# The profiler runs come last: a profiler that takes a process down must not take the other measurements with it.
#   gpurun --timeout 3600 -- 'bash tools/gpu/evidence.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
( timeout 900 python bench.py --config 1mm ) > $O/bench_1mm.json 2> $O/bench_1mm.err
( timeout 600 python bench.py --scene 0 --noise 1 --no-cpu-baseline --no-e2e --no-out-of-cache ) > $O/bench_4mm_round2_input.json 2> $O/bench_4mm_round2_input.err
( timeout 900 python bench.py --config partition ) > $O/bench_partition.json 2> $O/bench_partition.err
for hs in none gpu gpu-decimate full; do
  ( timeout 900 python bench.py --config scans --steps 12 --host-stage $hs ) > $O/bench_scans_$hs.json 2> $O/bench_scans_$hs.err
done
( timeout 900 python tools/e2e_bench.py --frames 5578 --out $O/e2e_5578.json ) > $O/e2e_5578.log 2>&1
( timeout 900 python tools/e2e_bench.py --frames 5578 --scene 0 --noise 1 --fuse-only --out $O/e2e_5578_round2_input.json ) > $O/e2e_5578_round2_input.log 2>&1
( timeout 900 python tools/e2e_bench.py --frames 5578 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color jpeg --fuse-only --out $O/e2e_colour_jpeg.json ) > $O/e2e_colour_jpeg.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
cd /tmp
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --repeats 1 --teardown > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/kt.txt 2>&1
python $R/tools/timeline.py $(find /tmp/kt -name "*.db" | head -1) > $O/timeline.txt 2>&1
# the front chain alone (everything on one stream): what the allocation kernel takes when nothing runs beside it
rm -rf /tmp/kt1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --no-profile --repeats 1 --tune overlap=0 --teardown > $O/kt_serial.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt1 -name "*.db" | head -1) > $O/kt_serial.txt 2>&1
# counters in their own passes (never together with a trace)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm_$n -o pmc -- python $R/bench.py --no-cpu-baseline --no-pmc --no-e2e --no-out-of-cache --no-single-frame --no-colour --repeats 1 --steps 320 --teardown > $O/pmc_$n.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pm_$n -name "*.db" | head -1) > $O/pmc_$n.txt 2>&1
done
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/evidence/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}; c = j.get("roofline_colour") or {}; s1 = j.get("roofline_single_frame") or {}
        print(f.split("/")[-1], j["value"], j["unit"], "| us/kernel", r.get("avg_kernel_us"), "frac", r.get("frac"), "hbm", r.get("hbm_frac"), "| colour", c.get("frames_per_s"), "| single:", s1.get("frames_per_s"), s1.get("avg_kernel_us"), s1.get("frac"), (s1.get("kernel_alone") or {}).get("frac"), "| idle", j.get("gpu_idle_pct"), (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json", ".err")).read()[-800:])
for f in sorted(glob.glob("gpurun_out/evidence/e2e*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], j["fuse"]["frames_per_s_end_to_end"], {k: j[k] for k in j if k.endswith("_s")})
PY
head -8 $O/kt.txt | cut -c1-150
