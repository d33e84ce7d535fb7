#!/bin/bash
# Round 4, at HEAD: the whole GPU suite, smoke, the two bench lines, the kernel trace of a scan end to end
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
( timeout 900 python tools/e2e_bench.py --frames 5578 --threads 4 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
cd /tmp
rm -rf /tmp/kti; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kti -o kt -- python $R/tools/e2e_bench.py --frames 2400 --fuse-only --threads 4 --out /tmp/e2e_kt.json > $O/kt_e2e.log 2>&1
DB=$(find /tmp/kti -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/kt_e2e.txt 2>&1
python $R/tools/timeline.py $DB -330 120 --skip k_synth_room > $O/timeline_e2e.txt 2>&1
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04final/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}
        e = j.get("end_to_end") or {}; e2 = j.get("end_to_end_rgbd") or {}
        print(f.split("/")[-1], j["value"], "depth-only", j.get("value_depth_only"), "| us", r.get("avg_kernel_us"), "frac", r.get("frac"), "| e2e", e.get("frames"), e.get("frames_per_s"), e.get("frames_per_s_first_and_second_run"),
              e.get("decode_threads"), (e.get("host_inflate") or {}).get("frames_per_s"), "| rgbd", e2.get("frames_per_s"), e2.get("frames_per_s_first_and_second_run"), "| parity", (j.get("parity") or {}).get("sha256_equal"))
    except Exception as ex:
        print(f, "ERR", ex); print(open(f.replace(".json", ".err")).read()[-600:])
j = json.load(open("gpurun_out/r04final/e2e_5578_gpu.json")); print("e2e tool", j["fuse"]["frames_per_s_end_to_end"], j["fuse"]["first_run_of_the_process"], {k: j[k] for k in j if k.endswith("_s")})
PY
head -9 $O/kt_e2e.txt | cut -c1-150
