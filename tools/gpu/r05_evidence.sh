#!/bin/bash
# Round 5 evidence: the whole GPU suite, smoke, the default bench line and the driver's command.   gpurun --timeout 2400 -- 'bash tools/gpu/r05_evidence.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ev
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1200 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
python - <<'PY'
import json
for f in ("bench_4mm", "bench_4mm_driver_args"):
    try:
        j = json.loads([l for l in open("gpurun_out/r05ev/%s.json" % f).read().splitlines() if l.startswith("{")][0])
        r = j["roofline"]; e = j.get("end_to_end") or {}; g = j.get("end_to_end_rgbd") or {}
        cal = r.get("valu_peak_calibration") or {}
        print(f, "value", j["value"], j.get("value_depth_only"), "| kernel us", r.get("avg_kernel_us"), "frac", r.get("frac"), "overlap floor", cal.get("frac_overlap_floor"), "old ratio", r.get("issue_ratio_4_cycles"),
              "cyc/inst", cal.get("cycles_per_instruction_and_simd_measured"), "| hbm", r.get("hbm_frac"), "traffic/alg", (r.get("traffic_detail") or {}).get("traffic_over_alg"))
        print("   ooc", json.dumps(r.get("hbm_out_of_cache"))[:900])
        print("   e2e", e.get("writer"), e.get("frames_per_s"), e.get("frames_per_s_first_and_second_run"), (e.get("other_writer") or {}).get("frames_per_s"), "| rgbd", g.get("frames_per_s"), g.get("frames_per_s_first_and_second_run"), g.get("decode_threads"))
        print("   single", (j.get("roofline_single_frame") or {}).get("frames_per_s"), (j.get("roofline_single_frame") or {}).get("frac"), "| cpu", (j.get("cpu_baseline") or {}).get("value"), "| parity", j.get("parity", {}).get("sha256_equal"), (j.get("parity_depth_only") or {}).get("sha256_equal"))
    except Exception as ex:
        print(f, "parse failed", ex)
PY
tail -2 $O/bench_4mm.err
