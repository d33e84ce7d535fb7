#!/bin/bash
# Round 4, closing run: what changed after tools/gpu/r04_evidence.sh (the device's inflate and JPEG entropy decoder in sf_fuse_run) -- the whole
# GPU suite, the bench lines whose end-to-end legs moved, the end-to-end tool on the 5 578-frame scan, the kernels alone and under the trace.
#   gpurun --timeout 2400 -- 'bash tools/gpu/r04_evidence2.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ev2
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_4mm.json 2> $O/bench_4mm.err
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_4mm_driver_args.json 2> $O/bench_4mm_driver_args.err
( timeout 900 python bench.py --config scans --steps 12 ) > $O/bench_scans_gpu.json 2> $O/bench_scans_gpu.err
( timeout 300 python tools/gpu/inflate_bench.py ) > $O/inflate_kernels.json 2> $O/inflate_kernels.err
( timeout 900 python tools/e2e_bench.py --frames 5578 --threads 4 --gpu-decimate --gpu-clean --out $O/e2e_5578_gpu.json ) > $O/e2e_5578_gpu.log 2>&1
( SF_INFLATE_HOST=1 timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --out $O/e2e_5578_host_inflate_16_threads.json ) > $O/e2e_5578_host_inflate_16_threads.log 2>&1
( SF_INFLATE_HOST=1 timeout 600 python tools/e2e_bench.py --frames 5578 --fuse-only --threads 4 --out $O/e2e_5578_host_inflate_4_threads.json ) > $O/e2e_5578_host_inflate_4_threads.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 3000 --color raw --fuse-only --out $O/e2e_colour_raw.json ) > $O/e2e_colour_raw.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --out $O/e2e_colour_jpeg_1296.json ) > $O/e2e_colour_jpeg_1296.log 2>&1
( SF_JPEG_GPU_HUFFMAN=1 timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --threads 4 --out $O/e2e_colour_jpeg_1296_gpu_huffman_4_threads.json ) > $O/e2e_colour_jpeg_1296_gpu_huffman_4_threads.log 2>&1
( timeout 600 python tools/e2e_bench.py --frames 2000 --color jpeg --color-res 1296x968 --fuse-only --threads 4 --out $O/e2e_colour_jpeg_1296_4_threads.json ) > $O/e2e_colour_jpeg_1296_4_threads.log 2>&1
cd /tmp
rm -rf /tmp/kti; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kti -o kt -- python $R/tools/e2e_bench.py --frames 2400 --fuse-only --threads 4 --out /tmp/e2e_kt.json > $O/kt_e2e.log 2>&1
DB=$(find /tmp/kti -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/kt_e2e.txt 2>&1
python $R/tools/timeline.py $DB 5400 110 > $O/timeline_e2e.txt 2>&1
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04ev2/bench*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
        r = j.get("roofline") or {}
        e = j.get("end_to_end") or {}; e2 = j.get("end_to_end_rgbd") or {}
        print(f.split("/")[-1], j["value"], j["unit"], "depth-only", j.get("value_depth_only"), "| us", r.get("avg_kernel_us"), "frac", r.get("frac"), "| e2e", e.get("frames_per_s"), e.get("frames_per_s_first_and_second_run"),
              e.get("decode_threads"), e.get("host_inflate"), "| rgbd", e2.get("frames_per_s"), e2.get("frames_per_s_first_and_second_run"), "| parity", (j.get("parity") or {}).get("sha256_equal"), "| idle", j.get("gpu_idle_pct"))
    except Exception as ex:
        print(f, "ERR", ex); print(open(f.replace(".json", ".err")).read()[-600:])
for f in sorted(glob.glob("gpurun_out/r04ev2/e2e*.json")):
    j = json.load(open(f)); print(f.split("/")[-1], j["fuse"]["frames_per_s_end_to_end"], j["fuse"]["first_run_of_the_process"]["frames_per_s_end_to_end"], j["fuse"]["decode_threads"], {k: j[k] for k in j if k.endswith("_s")})
print(open("gpurun_out/r04ev2/inflate_kernels.json").read())
PY
head -12 $O/kt_e2e.txt | cut -c1-150
