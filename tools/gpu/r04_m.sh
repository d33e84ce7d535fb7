#!/bin/bash
# Round 4, last call: the bench lines at HEAD (the end-to-end leg over the whole scan) and the timeline of a scan end to end
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
cd /tmp
rm -rf /tmp/kti; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kti -o kt -- python $R/tools/e2e_bench.py --frames 2400 --fuse-only --threads 4 --out /tmp/e2e_kt.json > $O/kt_e2e.log 2>&1
DB=$(find /tmp/kti -name "*.db" | head -1)
python $R/tools/timeline.py $DB -330 120 --skip k_synth_room > $O/timeline_e2e.txt 2>&1
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04m/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}
        e = j.get("end_to_end") or {}; e2 = j.get("end_to_end_rgbd") or {}
        print(f.split("/")[-1], j["value"], j["unit"], "depth-only", j.get("value_depth_only"), "| us", r.get("avg_kernel_us"), "frac", r.get("frac"), "| e2e", e.get("frames"), e.get("frames_per_s"), e.get("frames_per_s_first_and_second_run"),
              e.get("decode_threads"), e.get("host_inflate"), "| rgbd", e2.get("frames"), e2.get("frames_per_s"), e2.get("frames_per_s_first_and_second_run"), "| parity", (j.get("parity") or {}).get("sha256_equal"), "| mc", (e.get("marching_cubes") or {}).get("total"))
    except Exception as ex:
        print(f, "ERR", ex); print(open(f.replace(".json", ".err")).read()[-600:])
PY
tail -4 $O/bench_4mm.err; head -40 $O/timeline_e2e.txt | cut -c1-100
