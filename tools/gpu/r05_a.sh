#!/bin/bash
# Round 5, call A: the VALU issue ceiling (tools/gpu/valu_peak.hip: clocks + the bench's own counter ratio), the new GPU tests on reference-written
# streams, a bench line (end_to_end on the reference writer's file).   gpurun --timeout 1500 -- 'bash tools/gpu/r05_a.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/gpu/valu_peak.hip -o /tmp/valu_peak 2> $O/valu_peak_build.err
( timeout 300 /tmp/valu_peak 20000 ) > $O/valu_peak.json 2> $O/valu_peak.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05a/valu_peak.json"))
print(j["device"], j["cus"], j["clock_rate_khz"])
for s in j["streams"]:
    print("%-34s W=%d  inst/tick/SIMD %.4f (span %.4f)  ticks/inst %.3f  clock %.3g  census %s slots %d ms %.3f" % (s["op"][:34], s["waves_per_simd"], s["inst_per_tick_per_simd"], s["inst_per_tick_per_simd_span"], s["ticks_per_inst"], s["memtime_ticks_per_s"], s["waves_per_simd_census"], s["simd_slots_seen"], s["event_ms"]))
PY
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "reference or corrupt or inflate or fuse_run" ) > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/r05a/bench_4mm.json").read().splitlines() if l.startswith("{")][0])
    e = j["end_to_end"]; g = j.get("end_to_end_rgbd") or {}
    print("value", j["value"], j.get("value_depth_only"), "frac", j["roofline"]["frac"], j["roofline"]["avg_kernel_us"])
    print("e2e", e["writer"], e["frames_per_s"], e["frames_per_s_first_and_second_run"], e["compressed_bytes_per_frame"], e["depth_inflated_on_device"], e["depth_inflated_on_host"], "other", e["other_writer"], {k: v for k, v in (e["inflate_kernels"] or {}).items() if k != "what"}, e["host_inflate"])
    print("rgbd", g.get("writer"), g.get("frames_per_s"), g.get("frames_per_s_first_and_second_run"), g.get("jpeg_bytes_per_picture"), g.get("decode_threads"), g.get("decode_ms_per_frame_per_thread"))
    print("parity", j["parity"]["sha256_equal"])
except Exception as ex:
    print("bench parse failed", ex)
PY
tail -3 $O/bench_4mm.err
# counters last (a profiler that takes the process down must not take the rest with it): the bench's own ratio on streams of known issue rate
cd /tmp
rm -rf /tmp/pm_valu; timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_valu -o pmc -- /tmp/valu_peak 4000 > $O/pmc_valu_peak.log 2>&1
python - <<'PY' > $O/pmc_valu_peak.txt 2>&1
# per dispatch (launch order per kernel: W = 1, 1, 2, 2, 4, 4, 8, 8 -- warm-up then measurement): the ratio bench.py calls valu_util
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/pm_valu/**/*.db", recursive=True)[0])
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id"))
by = {}
for d, k, c, v in rows:
    by.setdefault((d, k), {})[c] = by.setdefault((d, k), {}).get(c, 0) + v
seen = {}
print("%-28s %2s %14s %14s %14s %14s %9s %9s" % ("kernel", "W", "ACTIVE_INST_VALU", "INSTS_VALU", "GUI_ACTIVE/8", "WAVE_CYCLES", "valu_util", "act/inst"))
for (d, k), c in sorted(by.items()):
    i = seen.get(k, 0); seen[k] = i + 1
    if i % 2 == 0:
        continue
    W = (1, 2, 4, 8)[(i // 2) % 4]
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    util = c.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * gui / 4.0) if gui else 0
    print("%-28s %2d %14.0f %14.0f %14.0f %14.0f %9.4f %9.4f" % (k[:28], W, c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_INSTS_VALU", 0), gui, c.get("SQ_WAVE_CYCLES", 0), util,
                                                        c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1)))
PY
cat $O/pmc_valu_peak.txt | cut -c1-200 | head -70
