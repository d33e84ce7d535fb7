#!/bin/bash
# Round 5, call B: the VALU issue table (tools/gpu/valu_peak.hip) -- clocks, then the bench's own counter ratio per stream.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O2 -std=c++20 --offload-arch=gfx950 tools/gpu/valu_peak.hip -o /tmp/valu_peak 2> $O/valu_peak_build.err
( timeout 300 /tmp/valu_peak 8000 ) > $O/valu_peak.json 2> $O/valu_peak.err
cd /tmp
rm -rf /tmp/pm_valu; timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_valu -o pmc -- /tmp/valu_peak 4000 > $O/pmc_valu_peak.log 2>&1
python - <<'PY' > $O/pmc_valu_peak.txt 2>&1
# per dispatch (launch order per kernel: W = 1, 1, 2, 2, 4, 4, 8, 8 -- warm-up then measurement): the ratio bench.py calls valu_util, and SIMD cycles per wave-instruction
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/pm_valu/**/*.db", recursive=True)[0])
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id"))
by = {}
for d, k, c, v in rows:
    e = by.setdefault((d, k), {})
    e[c] = e.get(c, 0) + v
seen = {}
print("%-46s %2s %14s %12s %12s %9s %9s %12s" % ("kernel", "W", "ACTIVE_INST_VALU", "INSTS_VALU", "GUI_ACTIVE/8", "valu_util", "act/inst", "cyc/inst/SIMD"))
for (d, k), c in sorted(by.items()):
    i = seen.get(k, 0); seen[k] = i + 1
    if i % 2 == 0:
        continue
    W = (1, 2, 4, 8)[(i // 2) % 4]
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    inst = c.get("SQ_INSTS_VALU", 0)
    util = c.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * gui / 4.0) if gui else 0
    name = k.split("(")[0].replace("void ", "").replace("k_", "", 1)
    print("%-46s %2d %14.0f %12.0f %12.0f %9.4f %9.4f %12.3f" % (name[:46], W, c.get("SQ_ACTIVE_INST_VALU", 0), inst, gui, util, c.get("SQ_ACTIVE_INST_VALU", 0) / max(inst, 1), gui * 1024 / max(inst, 1)))
PY
python - <<'PY'
import json
j = json.load(open("/root/repo/gpurun_out/r05b/valu_peak.json"))
w1 = {s["op"]: s for s in j["streams"] if s["waves_per_simd"] == 1}
for s in j["streams"]:
    clk = w1[s["op"]]["mean_wave_ticks"] / (w1[s["op"]]["event_ms"] * 1e-3)     # one wave per SIMD: the wave's ticks span the launch
    n = j["instructions_per_wave"] * (0.25 if s["op"] == "v_rcp_f32" else 1) * s["waves_per_simd"]
    print("%-34s W=%d event %.3f ms  cyc/inst/SIMD(event x %.2f GHz) %.3f   one wave: %.3f ticks/inst" % (s["op"][:34], s["waves_per_simd"], s["event_ms"], clk / 1e9, s["event_ms"] * 1e-3 * clk / n, s["mean_wave_ticks"] / (n / s["waves_per_simd"])))
PY
cat $O/pmc_valu_peak.txt | cut -c1-160
