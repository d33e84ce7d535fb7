#!/usr/bin/env python3
"""bench.py -- RGB-D frames/s integrated on MI355X + the roofline of the integrate kernel.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 4mm|1mm|scans|partition]      (N > 1 without a launcher: bench.py re-executes itself
                                                                                               under torch.distributed.run with N ranks, or refuses)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

--config (BASELINE.json `configs`; the default is the one the metric is quoted on):
  4mm        configs[1]  scene0000_00-scale synthetic stream (5 578 frames, 640x480, 4 mm voxels, 2^19 hash buckets).  One step = one RGB-D
                         frame -- a 640x480 u16 depth image AND a 640x480 RGB image -- through the whole per-frame hot path (pre-pass, block
                         allocation, frustum compaction, TSDF + colour integrate); both streams are resident in HBM before the timed region;
                         with N > 1 every rank fuses its own scan (weak scaling).  --depth-only times the geometry-only path (rounds 1-3's
                         headline; reported beside the metric as value_depth_only in every default line).
  1mm        configs[2]  the same stream at 1 mm voxels / 2^22 buckets / 2^25 SDF blocks (137 GB of tiles): ~1.4 M tiles = 5.9 GB touched
                         per frame, far beyond the 256 MiB Infinity Cache -- the out-of-cache HBM roofline of the one-frame kernel.
  scans      configs[3]  independent scans (room size +-20 %, 300..6000 frames) popped longest-first from one queue by all ranks, no
                         collective; one step = one scan through fusion + marching cubes (+ the host stage: clean, decimate x 2, segment).
  partition  configs[4]  ONE long scan (default 50 000 frames through a 10-room corridor world) with the block space dealt in stripes
                         to the ranks, boundary layers all-gathered over RCCL before marching cubes; one step = one frame (strong scaling).

`roofline` describes the dominant kernel of the timed region and every `frac` is a fraction of a hardware peak (<= 1):
  * 32 frames per launch (the default schedule) is VALU-issue bound: bound = "valu", frac = SQ_INSTS_VALU of the launch (separate
    rocprofv3 --pmc pass) x the mean issue cost of the kernel's frame loop at the per-class costs MEASURED on the MI355X
    (tools/gpu/valu_peak.hip, tools/valu_cost_model.py: roofline.valu_peak_calibration) over the SIMD cycles of the launch;
    issue_ratio_4_cycles = rounds 2-4's SQ_ACTIVE_INST_VALU x 4 / SIMD cycles, kept for continuity (it is not a utilisation: 1.8 on a
    pure v_fma stream); hbm_frac = counter traffic / time / 8 TB/s, alg_equiv_GBs = what the launch would have moved without temporal
    blocking (SURVEY 8d algorithmic bytes / time: NOT a roofline fraction).
  * `roofline_single_frame` (one frame per launch, what a live stream gets) is HBM bound: achieved = algorithmic bytes / time.
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# before anything initialises HIP (torch.cuda does, long before libscanfuse.so is loaded): sf_fuse_run drives seven streams, the runtime's
# default of four hardware queues puts some of them in one queue (scannet_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

TOTAL_FRAMES = 5578
W, H = 640, 480
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s is what a float4 copy reaches)
NUM_SIMDS = 1024        # 256 CUs x 4 SIMDs
MALL_BYTES = 256 << 20  # Infinity Cache: a per-frame tile footprint below this is served on-die between launches

TUNE = {}   # --tune key=value switches (scanfuse_internal.h), applied to every fuser

CONFIGS = {
    "4mm": dict(voxel_size=0.004, hash_num_buckets=1 << 19, num_sdf_blocks=1 << 20, steps=TOTAL_FRAMES - 64, warmup=64,
                label="configs[1]: scene0000_00-scale synthetic stream (5578-frame box-room walk, 640x480 u16 depth, 4 mm voxels, 2^19 hash buckets x 10, 2^20 SDF blocks)"),
    "1mm": dict(voxel_size=0.001, hash_num_buckets=1 << 22, num_sdf_blocks=1 << 25, steps=192, warmup=16,
                label="configs[2]: the same stream at 1 mm voxels, 2^22 hash buckets x 10, 2^25 SDF blocks (137 GB of tiles reserved)"),
}


# ---------------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle port on a bounded sample at 1, 8 and all usable threads; the reference's own SensReader decode
# beside ours; the parity of the GPU path with the oracle on exactly the frames the CPU leg fused
# ---------------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(depth_host, poses, voxel, frames=200, rgb_host=None):
    """the CPU path = reference SensorData decode (compiled from /root/reference when this repo was built) + the oracle
    (our CPU port of the same specification, OpenMP over blocks) on the first `frames` frames of the same stream, at 1, 8 and all the threads this
    container may use.  Returns (dict for the JSON line, the oracle volume of the all-threads run for the parity leg)."""
    from oracle import oracle as orc
    from scannet_amd import _abi
    allt = _abi.usable_cpus()   # the cgroup quota, not the 256 logical CPUs the container shows
    n = min(frames, len(depth_host))
    decode = reference_decode_ms(depth_host[:32], poses[:32])
    t_dec = (decode or {}).get("reference_sensreader")
    by, keep = {}, None
    n_all = n
    for th in sorted({1, min(8, allt), allt}):
        vol = orc.Volume(orc.default_params(W, H, voxel), threads=th)
        budget = 30.0 if th == 1 else 12.0   # seconds: the 4 mm stream finishes its 200 frames inside these, 1 mm voxels (0.25 s per frame on 16 threads) do not
        t0 = time.perf_counter()
        done = 0
        for i in range(n):
            vol.integrate(depth_host[i], poses[i], rgb=None if rgb_host is None else rgb_host[i])
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        e = {"frames": done, "frames_per_s_integrate": round(done / dt, 3), "seconds": round(dt, 2)}
        if t_dec:   # the reference decodes one frame per call on the caller's thread; `th` threads decode `th` frames at once
            e["frames_per_s_with_reference_decode"] = round(1.0 / (dt / done + t_dec * 1e-3 / th), 3)
        by[str(th)] = e
        if th == allt:
            keep, n_all = vol, done
        else:
            vol.close()
    n = n_all
    best = by[str(allt)]
    out = {"value": best.get("frames_per_s_with_reference_decode", best["frames_per_s_integrate"]), "unit": "frames/s", "cores": allt, "kind": "port",
           "by_threads": by, "reference_decode": decode,
           "cpu_model": _cpu_model(), "nproc_visible": os.cpu_count(),
           "colour": rgb_host is not None,
           "sample": "first %d %s frames of the same stream (fewer where a leg ran into its time budget: by_threads.*.frames): reference SensorData depth decode (oracle/_ref/libref_sens.so, -O2, one frame per thread) + "
                     "oracle/tsdf_oracle.c (-O2 -fopenmp, blocks over threads) at 1 / 8 / %d threads (= the CPUs this container may use, of %d visible); "
                     "`value` is the all-threads figure" % (n, "RGB-D (depth + resident raw colour)" if rgb_host is not None else "depth-only", allt, os.cpu_count() or allt)}
    return out, keep, n


def reference_decode_ms(depth_host, poses):
    """The part of the path the reference DOES have a CPU implementation of: per-frame depth decode (SensReader, compiled from the
    reference's sources into oracle/_ref/libref_sens.so when this repo was built) beside this library's decoder, one core each,
    on a .sens written from the first frames of the stream.  None when the reference build is not there."""
    try:
        from oracle import oracle as orc
        from scannet_amd import _abi, sens, synth
        if not orc.ref_sens_available():
            return None
        d = tempfile.mkdtemp(prefix="sf_refdec_", dir="/tmp")
        path = os.path.join(d, "s.sens")
        K = synth.intrinsic_matrix(W, H)
        sd = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=1)
        for i in range(len(depth_host)):
            sd.add_frame(depth_host[i], poses[i].reshape(4, 4))
        sd.save(path)
        sd.close()
        out = np.zeros((H, W), np.uint16)
        R = orc.ref_sens()
        h = R.ref_sens_open(path.encode())
        t0 = time.perf_counter()
        for i in range(len(depth_host)):
            R.ref_sens_decode_depth(h, i, out.ctypes.data_as(C.c_void_p))
        t_ref = (time.perf_counter() - t0) / len(depth_host)
        R.ref_sens_close(h)
        L = _abi.lib()
        s2 = sens.SensorData(path)
        t0 = time.perf_counter()
        for i in range(len(depth_host)):
            L.sf_sens_decode_depth(s2._h, C.c_uint64(i), out.ctypes.data_as(C.c_void_p))
        t_ours = (time.perf_counter() - t0) / len(depth_host)
        shutil.rmtree(d, ignore_errors=True)
        return {"unit": "ms per 640x480 depth frame, one core", "reference_sensreader": round(t_ref * 1e3, 3), "this_library": round(t_ours * 1e3, 3),
                "kind": "reference", "sample": "%d frames, zlib depth, oracle/_ref/libref_sens.so (the reference's sensorData.h + stb, -O2)" % len(depth_host)}
    except Exception:
        return None



def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def parity_leg(ovol, n, frames_dev, stride, poses, params, local_rank, rgb_dev=None):
    """The same n frames the CPU leg fused, through the HIP path of THIS run's build: block set and every voxel byte (sdf, r, g, b, weight) must
    be identical."""
    import hashlib
    from scannet_amd import fusion
    oc, ov = ovol.export()
    with fusion.Fuser(params, device=local_rank, **TUNE) as f:
        f.integrate_batch_device(frames_dev.data_ptr(), stride, poses[:n], None if rgb_dev is None else rgb_dev.data_ptr(), W * H * 3)
        gc, gv = f.export_blocks()
        fails = f.stats()["alloc_failures"]
    ho = hashlib.sha256(oc.tobytes() + ov.tobytes()).hexdigest()
    hg = hashlib.sha256(gc.tobytes() + gv.tobytes()).hexdigest()
    return {"frames": n, "blocks": int(len(gc)), "blocks_oracle": int(len(oc)), "sha256_equal": ho == hg, "sha256": hg[:16], "alloc_failures": fails,
            "weight_max_seen": int(gv["w"].max()) if len(gc) else 0, "colour": rgb_dev is not None,
            "coloured_voxels": int(((gv["r"] | gv["g"] | gv["b"]) > 0).sum()) if len(gc) else 0,
            "what": "sha256 over (block coordinates sorted by x, y, z; 512 x 8-byte voxels {f32 sdf, u8 r, g, b, weight} per block) of oracle/tsdf_oracle.c and of the HIP path "
                    "(the default schedule: 32 frames per pass) after the first %d %s frames of this run's stream" % (n, "RGB-D" if rgb_dev is not None else "depth-only")}


# ---------------------------------------------------------------------------------------------------------------------------------------
# PMC passes: this script re-run as a child under rocprofv3 --pmc (no trace domains), per-launch averages of the integrate kernel
# ---------------------------------------------------------------------------------------------------------------------------------------
def pmc_pass(args, counters, config, steps, warmup, single_frame, depth_only, timeout_s=420):
    """One rocprofv3 --pmc pass over the first `steps` timed frames of this same script.  Returns ({counter: average per integrate launch of
    the timed region}, the child's JSON line) or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="sf_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--pmc"] + list(counters) + ["-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--config", config, "--steps", str(steps),
               "--warmup", str(warmup), "--child", "--no-profile", "--teardown", "--scene", str(args.scene), "--noise", str(args.noise)] + (["--single-frame"] if single_frame else []) + \
              (["--depth-only"] if depth_only else []) + \
              sum((["--tune", "%s=%d" % kv] for kv in TUNE.items()), [])
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None
        child = None
        for ln in r.stdout.splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                child = json.loads(ln)
        if child is None:
            return None
        launches = child["config"]["integrate_launches"]
        # one frame per launch runs the software-pipelined k_integrate_pipe, batches run k_integrate; the integrate launches of the timed
        # region are the LAST `launches` dispatches of the kernel (warm-up comes first)
        pat = "%k_integrate_pipe%" if (single_frame and depth_only) else ("%k_integrate<1, 0,%" if depth_only else "%k_integrate<1, 2,%")   # <SIGN, COLOR (0 none, 2 colour), ...>
        db = sqlite3.connect(dbs[0])
        out = {}
        for c in counters:
            rows = [v for (v,) in db.execute("select value from counters_collection where counter_name = ? and kernel_name like ? order by dispatch_id", (c, pat))]
            if launches <= 0 or len(rows) < launches:
                db.close()
                return None
            out[c] = sum(rows[-launches:]) / launches
        # the kernels in FRONT of the integrate launch (one launch each per pass): the same counters from the same pass, and their duration under
        # the profiler (kernels serialised: what each takes alone)
        front = {}
        for key, fp in FRONT_KERNELS.items():
            e = {}
            for c in counters:
                rows = [v for (v,) in db.execute("select value from counters_collection where counter_name = ? and kernel_name like ? order by dispatch_id", (c, fp))]
                if len(rows) >= launches:
                    e[c] = sum(rows[-launches:]) / launches
            try:
                dur = [t for (t,) in db.execute("select end - start from kernels where name like ? order by start", (fp,))]
                if len(dur) >= launches:
                    e["avg_us_alone"] = round(sum(dur[-launches:]) / launches / 1e3, 2)
            except sqlite3.Error:
                pass
            if e:
                front[key] = e
        db.close()
        out["_front"] = front
        return out, child
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


FRONT_KERNELS = {"k_alloc_ray": "%k_alloc_ray%", "k_compactify": "%k_compactify%", "k_prepass": "%k_prepass%"}


def pmc_traffic(args, config, steps, warmup, single_frame, depth_only):
    """HBM-side traffic of the integrate kernel per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE passes.  Corrections as
    MI355X_MICROARCH.md (HBM) prescribes and tools/pmc_calibrate.py confirmed for this kernel's 16 B/lane pattern
    (profiles/r01_b_alloc_bitmap_rocprofv3.txt): both counters are KiB per dispatch, FETCH_SIZE reports exactly half of the bytes read,
    WRITE_SIZE the bytes written.  Infinity-Cache hits are counted (fabric-side counters), so this is an upper bound on DRAM traffic."""
    a = pmc_pass(args, ["FETCH_SIZE"], config, steps, warmup, single_frame, depth_only)
    b = pmc_pass(args, ["WRITE_SIZE"], config, steps, warmup, single_frame, depth_only) if a else None
    if not a or not b:
        return None
    read_b, write_b = 2.0 * a[0]["FETCH_SIZE"] * 1024.0, b[0]["WRITE_SIZE"] * 1024.0
    alg = a[1]["config"].get("alg_bytes_per_launch")
    front = {}
    for k in FRONT_KERNELS:
        fa, fb = a[0].get("_front", {}).get(k, {}), b[0].get("_front", {}).get(k, {})
        if "FETCH_SIZE" in fa and "WRITE_SIZE" in fb:
            front[k] = {"read_bytes": round(2.0 * fa["FETCH_SIZE"] * 1024.0), "write_bytes": round(fb["WRITE_SIZE"] * 1024.0), "avg_us_alone": fa.get("avg_us_alone")}
            if fa.get("avg_us_alone"):
                front[k]["hbm_frac_alone"] = round((front[k]["read_bytes"] + front[k]["write_bytes"]) / (fa["avg_us_alone"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    return {"bytes": round(read_b + write_b), "read_bytes": round(read_b), "write_bytes": round(write_b), "front_chain": front,
            "sample": "integrate launches of frames %d..%d, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 "
                      "(gfx950 128-B requests tallied at 64 B), WRITE_SIZE x1, KiB -> bytes; Infinity-Cache hits are counted" % (warmup, warmup + steps - 1),
            "alg_bytes_same_launches": alg, "traffic_over_alg": round((read_b + write_b) / alg, 4) if alg else None}


def pmc_mempipe(args, config, steps, warmup, single_frame, depth_only):
    """The memory-pipe side of the integrate launch (the kernel is co-bound by the texture addresser / L1): TA_TA_BUSY (cycles a
    texture addresser is busy, summed over the 256 CUs), TCP_TOTAL_CACHE_ACCESSES (L1 tag look-ups, summed) and TA_BUFFER_WAVEFRONTS (buffer-gather wave
    instructions), each over the CU cycles of the launch (256 CUs x GRBM_GUI_ACTIVE per XCD)."""
    a = pmc_pass(args, ["TA_TA_BUSY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TA_BUFFER_WAVEFRONTS_sum", "GRBM_GUI_ACTIVE"], config, steps, warmup, single_frame, depth_only)
    if not a:
        return None
    v = a[0]
    cu_cycles = 256.0 * v["GRBM_GUI_ACTIVE"] / 8.0
    if cu_cycles <= 0:
        return None
    return {"ta_busy": round(v["TA_TA_BUSY_sum"] / cu_cycles, 4), "tcp_tag_lookups_per_cu_clk": round(v["TCP_TOTAL_CACHE_ACCESSES_sum"] / cu_cycles, 4),
            "tcp_tag_lookups_per_gather": round(v["TCP_TOTAL_CACHE_ACCESSES_sum"] / v["TA_BUFFER_WAVEFRONTS_sum"], 2) if v["TA_BUFFER_WAVEFRONTS_sum"] else None,
            "ta_busy_cycles": round(v["TA_TA_BUSY_sum"]), "tcp_total_cache_accesses": round(v["TCP_TOTAL_CACHE_ACCESSES_sum"]),
            "ta_buffer_wavefronts": round(v["TA_BUFFER_WAVEFRONTS_sum"]), "gui_active_clocks_per_xcd": round(v["GRBM_GUI_ACTIVE"] / 8.0)}


_COST_MODEL = {}


def valu_cost_model(depth_only):
    """tools/valu_cost_model.py on the library this process loaded: the frame loop of the timed integrate kernel priced with the per-class issue costs
    tools/gpu/valu_peak.hip measured on the MI355X (profiles/r05_valu_issue_table.txt).  None when the disassembler is not there."""
    # <SIGN 1, COLOR, TAB, WM 2, ROWS false, NJ 4, XR>: the variant every pass but a call's last runs (XR: the x-row lane layout, tune "xrow", default on)
    key = "k_integrateILi1ELi%dELb1ELi2ELb0ELi4ELb%dE" % (0 if depth_only else 2, 1 if TUNE.get("xrow", 1) else 0)
    if key not in _COST_MODEL:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("valu_cost_model", os.path.join(ROOT, "tools", "valu_cost_model.py"))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            text = m.disassemble(os.path.join(ROOT, "scannet_amd", "libscanfuse.so"))
            _, lines = m.kernel_body(text, key)
            e = m.price(m.frame_loop(lines))
            e["costs_cycles"] = {"fast": m.C_FAST, "slow": m.C_SLOW, "trans": m.C_TRANS}
            e.pop("top_opcodes", None)
            _COST_MODEL[key] = e
        except BaseException:   # SystemExit of the tool included: the roofline then falls back to the uncalibrated ratio and says so
            _COST_MODEL[key] = None
    return _COST_MODEL[key]


def pmc_valu(args, config, steps, warmup, single_frame, depth_only):
    """VALU issue utilisation of the integrate kernel: SQ_ACTIVE_INST_VALU (quad-cycles a SIMD spends issuing VALU, summed over SIMDs)
    over the SIMD quad-cycles of the launch = 1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) / 4 -- the method of profiles/r01_c."""
    a = pmc_pass(args, ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"], config, steps, warmup, single_frame, depth_only)
    if not a:
        return None
    v, child = a
    slots = NUM_SIMDS * (v["GRBM_GUI_ACTIVE"] / 8.0) / 4.0
    blk = child["roofline_inputs"]["block_frames_per_launch"] if "roofline_inputs" in child else None
    front = {}
    for k, e in v.get("_front", {}).items():
        if all(c in e for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")) and e["GRBM_GUI_ACTIVE"] > 0:
            front[k] = {"valu_util_alone": round(e["SQ_ACTIVE_INST_VALU"] / (NUM_SIMDS * (e["GRBM_GUI_ACTIVE"] / 8.0) / 4.0), 4), "insts_valu": round(e["SQ_INSTS_VALU"]),
                        "valu_frac_2cycle_alone": round(e["SQ_INSTS_VALU"] * 2.0 / (NUM_SIMDS * (e["GRBM_GUI_ACTIVE"] / 8.0)), 4),
                        "avg_us_alone": e.get("avg_us_alone"), "valu_insts_share_of_the_pass": round(e["SQ_INSTS_VALU"] / (e["SQ_INSTS_VALU"] + v["SQ_INSTS_VALU"]), 4)}
    cm = None if single_frame else valu_cost_model(depth_only)
    simd_cycles = NUM_SIMDS * (v["GRBM_GUI_ACTIVE"] / 8.0)
    cal = None
    if cm and simd_cycles:
        cal = {"frac_serial": round(v["SQ_INSTS_VALU"] * cm["cycles_serial_per_instruction"] / simd_cycles, 4),
               "frac_overlap_floor": round(v["SQ_INSTS_VALU"] * cm["cycles_overlapped_per_instruction"] / simd_cycles, 4),
               "cycles_per_instruction_and_simd_measured": round(simd_cycles / v["SQ_INSTS_VALU"], 3),
               "frame_loop": cm, "issue_table": "profiles/r05_valu_issue_table.txt (tools/gpu/valu_peak.hip)",
               "what": "SQ_INSTS_VALU of the launch x the mean issue cost of the kernel's frame loop (its instructions priced by class: fp32 / simple integer with vector "
                       "sources 2.25 cycles per wave-instruction and SIMD, v_pk_*, conversions, compares, selects, scalar-source forms, three-operand integer 4.25, "
                       "transcendentals 8.3 -- measured on this part) / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs).  frac_serial prices every instruction as if nothing "
                       "overlapped; frac_overlap_floor lets the fp32 fast class issue beside the other classes, as alternating streams do in the microbenchmark"}
    return {"front_chain": front, "calibrated": cal,
            "valu_frac_2cycle": min(1.0, round(v["SQ_INSTS_VALU"] * 2.0 / simd_cycles, 4)) if simd_cycles else None,
            "valu_util": round(v["SQ_ACTIVE_INST_VALU"] / slots, 4) if slots else None, "active_inst_valu": round(v["SQ_ACTIVE_INST_VALU"]),
            "insts_valu": round(v["SQ_INSTS_VALU"]), "gui_active_clocks_per_xcd": round(v["GRBM_GUI_ACTIVE"] / 8.0),
            "valu_insts_per_voxel_frame": round(v["SQ_INSTS_VALU"] * 64.0 / (blk * 512.0), 2) if blk else None,
            "sample": "rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE (own pass, kernels serialised by the profiler), integrate launches of frames %d..%d"
                      % (warmup, warmup + steps - 1)}


# ---------------------------------------------------------------------------------------------------------------------------------------
# configs[1] / configs[2]: one resident stream per rank
# ---------------------------------------------------------------------------------------------------------------------------------------
def end_to_end(frames_dev, poses, n, params, local_rank, torch, colour=None, cache=None, alt_frames=1024):
    """SURVEY 8d "End-to-end frames/s": the first n frames of the run's stream in a .sens in /tmp, then sf_fuse_run: file -> host threads -> pinned
    ring -> H2D -> inflate on the GPU -> fusion.  Wall time from the first byte read to the last kernel.  The file is written by the REFERENCE
    writer when oracle/_ref/libref_sens.so is there (SensorData::createFrame per frame: stb deflate at quality 8, sensorData.h:659-670 ->
    stb_image_write.h:721-823 -- the bytes real ScanNet files hold; outside the timed region, as every write is); the same frames through this
    library's writer are timed beside it (`other_writer`).  `frames_per_s` is the FIRST run of the file in this process (one process per scan is the
    pipeline's contract, Server/scan_processor.py:138); the second run is in `frames_per_s_first_and_second_run`.
    colour = "jpeg1296": every frame also carries a baseline-JPEG colour picture at ScanNet's real 1296x968 with its own intrinsics, ~200 KB each
    (synth.textured_pictures; eight distinct ones cycled): entropy decoding on the host threads or the GPU, IDCT / upsampling / YCbCr->RGB on the GPU,
    then the colour pre-pass samples it under each depth pixel's ray.  The reference cannot encode JPEG off Windows (sensorData.h:576-593), so that
    container is assembled by this library from the reference writer's depth blobs and this library's JPEG encoder."""
    from scannet_amd import fusion, sens, synth
    d = tempfile.mkdtemp(prefix="sf_e2e_", dir="/tmp")
    try:
        host = frames_dev[:n].cpu().numpy().view(np.uint16)
        K = synth.intrinsic_matrix(W, H)
        P44 = poses[:n].reshape(-1, 4, 4)
        orc = None
        try:
            from oracle import oracle as _orc   # the reference WRITER only (input preparation, outside every timed region)
            if _orc.ref_sens_available() and hasattr(_orc.ref_sens(), "ref_sens_add_frames_mt"):
                orc = _orc
        except Exception:
            orc = None
        files = []   # (writer, path, seconds to write)
        prm = params
        if colour == "jpeg1296":
            from scannet_amd import calibrate
            cw, ch = 1296, 968
            KC = synth.intrinsic_matrix(cw, ch)
            blobs = [calibrate.jpeg_encode(img, 90, True) for img in synth.textured_pictures(cw, ch)]
            t0 = time.perf_counter()
            zd = orc.ref_write_sens(None, host, P44, K, want_blobs=True) if orc else None
            if cache is not None and zd is not None:
                cache["zd"] = zd   # the reference writer's depth streams of these frames: the depth-only leg stores the same blobs
            sd = sens.SensorData.create(cw, ch, W, H, KC, K, color_compression=2, depth_compression=1, sensor_name="StructureSensor")
            for i in range(n):
                if zd is not None:
                    sd.add_frame_blobs(zd[i], P44[i], color_blob=blobs[i % 8], timestamp_depth=33333 * i)
                else:
                    sd.add_frame(host[i], P44[i], color=blobs[i % 8], timestamp_depth=33333 * i)
            path = os.path.join(d, "rgbd.sens")
            sd.save(path)
            sd.close()
            del zd
            files.append(("depth: reference (stb); colour: this library's baseline-JPEG encoder" if orc else "this library", path, time.perf_counter() - t0))
            prm = type(params).from_buffer_copy(params)
            prm.color_width, prm.color_height = cw, ch
            prm.cfx, prm.cfy, prm.cmx, prm.cmy = synth.intrinsics(cw, ch)
            jpeg_bytes = round(sum(len(b) for b in blobs) / len(blobs))
        else:
            jpeg_bytes = None
            zd = (cache or {}).get("zd")
            if zd is not None and len(zd) >= n:   # the reference writer's streams of the same frames, compressed once for the RGB-D leg
                t0 = time.perf_counter()
                sd = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=1, sensor_name="StructureSensor")
                for i in range(n):
                    sd.add_frame_blobs(zd[i], P44[i], timestamp_depth=33333 * i)
                path = os.path.join(d, "reference_streams.sens")
                sd.save(path)
                sd.close()
                files.append(("reference (stb)", path, time.perf_counter() - t0))
            elif orc:
                try:
                    t0 = time.perf_counter()
                    path = os.path.join(d, "reference_writer.sens")
                    orc.ref_write_sens(path, host, P44, K)
                    files.append(("reference (stb)", path, time.perf_counter() - t0))
                except Exception:   # the leg then runs on this library's file alone
                    files = []
            if not files or shutil.disk_usage(d).free > 3 * n * W * H:   # room left for a second file of the scan (each ~0.6 of the pixels' bytes; 1.5 asked for)
                t0 = time.perf_counter()
                sd = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=1, sensor_name="StructureSensor")
                sd.add_depth_frames(host, P44)
                path = os.path.join(d, "this_library.sens")
                sd.save(path)
                sd.close()
                files.append(("this library", path, time.perf_counter() - t0))
        writer, path, t_write = files[0]
        size = os.path.getsize(path)
        sd = sens.SensorData(path)
        best = None
        first_run = None
        mc = None
        runs = []
        threads = min(4, os.cpu_count() or 4)   # the depth frames are inflated on the GPU: a host thread copies 330-370 KB per frame
        fusion.Fuser.prepare_run(sd, prm, local_rank)   # as bin/depthsensing does: the run's streams and rings for THIS file start being made before the fuser is
        for _ in range(2):   # the second pass reads the file from the page cache, as the stage after `convert` does, and finds the process's streams and pinned pool made
            with fusion.Fuser(prm, device=local_rank, **TUNE) as f:
                rs = f.run(sd, decode_threads=threads)   # JPEG colour too: the pictures are entropy-decoded on the device, a host thread only prepares the segment
                st = f.stats()
                runs.append(round(rs["frames_total"] / rs["seconds_total"], 1))
                if colour is None:   # what follows the fusion in the `improve` stage: marching cubes over the fused volume (second call: warm)
                    del_mesh = f.extract_mesh()
                    del del_mesh
                    del_mesh = f.extract_mesh()
                    del del_mesh
                    mc = f.mc_timing()
            if first_run is None:
                first_run = (rs, st)
            if best is None or rs["seconds_total"] < best[0]["seconds_total"]:
                best = (rs, st)
        alternatives = None
        if colour is not None:   # where the entropy decoding runs and on how many host threads: the same file, one run each (streams and pool exist by now)
            alternatives = {}
            for label, env, nt in (("device_huffman_4_threads", {"SF_JPEG_GPU_HUFFMAN": "1"}, threads), ("device_huffman_2_threads", {"SF_JPEG_GPU_HUFFMAN": "1"}, 2),
                                   ("host_huffman_all_threads", {"SF_JPEG_HOST_HUFFMAN": "1"}, 0), ("host_huffman_4_threads", {"SF_JPEG_HOST_HUFFMAN": "1"}, threads)):
                saved = {k: os.environ.get(k) for k in ("SF_JPEG_GPU_HUFFMAN", "SF_JPEG_HOST_HUFFMAN")}
                for k in saved:
                    os.environ.pop(k, None)
                os.environ.update(env)
                try:
                    with fusion.Fuser(prm, device=local_rank, **TUNE) as f:
                        ra = f.run(sd, 0, min(n, alt_frames), decode_threads=nt)   # a prefix: these are comparisons, not the leg's figure
                        alternatives[label] = {"frames_per_s": round(ra["frames_total"] / ra["seconds_total"], 1), "frames": int(ra["frames_total"]), "decode_threads": int(ra["decode_threads"]),
                                               "jpeg_entropy_on_device": int(ra.get("jpeg_entropy_on_device", -1)),
                                               "host_ms_per_frame_per_thread": round(1e3 * ra["seconds_decode_cpu"] / max(ra["frames_total"], 1), 3)}
                except Exception as ex:
                    alternatives[label] = {"error": str(ex)[:160]}
                finally:
                    for k, v in saved.items():
                        os.environ.pop(k, None)
                        if v is not None:
                            os.environ[k] = v
        other = None
        if len(files) > 1:   # the same frames through the other writer's streams
            w2, p2, tw2 = files[1]
            sd2 = sens.SensorData(p2)
            with fusion.Fuser(prm, device=local_rank, **TUNE) as f:
                r2 = f.run(sd2, decode_threads=threads)
                other = {"writer": w2, "frames_per_s": round(r2["frames_total"] / r2["seconds_total"], 1), "compressed_bytes_per_frame": round(os.path.getsize(p2) / n),
                         "depth_inflated_on_device": int(r2.get("depth_inflated_on_device", -1)), "write_s": round(tw2, 2), "run": "third of the process"}
            sd2.close()
        host_inflate = None
        kernels = None
        if colour is None:   # the two inflate kernels alone: 32 frames of this file per launch, HIP events around each (scanfuse_internal.h)
            try:
                from scannet_amd import _abi
                L = _abi.lib()
                L.sf_zlib_inflate_gpu_bench.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
                fr = sd.frames
                blobs = [np.frombuffer(fr[i * max(1, n // 32)].depth_compressed, np.uint8) for i in range(min(32, n))]   # the file's own streams, as its writer made them
                ptrs = (C.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
                sizes = (C.c_uint64 * len(blobs))(*[b.size for b in blobs])
                tt, tc = C.c_double(0), C.c_double(0)
                if L.sf_zlib_inflate_gpu_bench(ptrs, sizes, len(blobs), W * H * 2, local_rank, 10, 0, C.byref(tt), C.byref(tc)) == 0:
                    comp = sum(b.size for b in blobs)
                    kernels = {"frames_per_launch": len(blobs), "k_inflate_tokens_us": round(tt.value, 1), "k_inflate_copy_us": round(tc.value, 1),
                               "compressed_GBs_tokens": round(comp / (tt.value * 1e-6) / 1e9, 2), "output_GBs_copy": round(len(blobs) * W * H * 2 / (tc.value * 1e-6) / 1e9, 2),
                               "what": "csrc/inflate_gpu.hip alone on 32 frames of this file, resident: 1024 lanes per frame tokenise (speculative chunk starts, prefix sums, the per-byte plan), "
                                       "a 256-lane workgroup per frame makes the copies in 1024-byte groups with the window in LDS; counters: profiles/r04_pmc_inflate_kernels.txt",
                               "streams_of": writer}
            except Exception as ex:   # the leg is a measurement beside the metric: never take the line down
                kernels = {"error": str(ex)[:200]}
        if colour is None:   # rounds 1-3: every host thread this process may use inflates (SF_INFLATE_HOST)
            os.environ["SF_INFLATE_HOST"] = "1"
            try:
                with fusion.Fuser(prm, device=local_rank, **TUNE) as f:
                    rh = f.run(sd)
                    host_inflate = {"frames_per_s": round(rh["frames_total"] / rh["seconds_total"], 1), "decode_threads": int(rh["decode_threads"]),
                                    "decode_ms_per_frame_per_thread": round(1e3 * rh["seconds_decode_cpu"] / max(rh["frames_total"], 1), 3)}
            finally:
                del os.environ["SF_INFLATE_HOST"]
        sd.close()
        rs, st = best
        if mc is not None and mc["blocks"]:
            tile_b = mc["blocks"] * 729 * 8          # the 9^3 tile {sdf, rgbw} staged per block (8^3 own voxels + the +1 halo from 7 neighbours), per pass
            soup_b = mc["triangles"] * (8 + 3 * (8 + 12 + 4))
            out_b = mc["vertices"] * (12 + 4 + 8) + mc["triangles"] * (12 + 8)
            mc["count_pass_GBs"] = round(tile_b / (mc["count_pass"] * 1e-3) / 1e9, 1) if mc["count_pass"] > 0 else None
            mc["count_pass_hbm_frac"] = round(tile_b / (mc["count_pass"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mc["count_pass"] > 0 else None
            mc["emit_pass_GBs"] = round((tile_b + soup_b) / (mc["scan_emit_pass"] * 1e-3) / 1e9, 1) if mc["scan_emit_pass"] > 0 else None
            mc["download_bytes"] = out_b
            mc["download_GBs"] = round(out_b / (mc["downloads_device"] * 1e-3) / 1e9, 1) if mc["downloads_device"] > 0 else None
            mc["what"] = ("sf_fuser_extract_mesh over the %d live blocks of the fused prefix, second call, milliseconds per phase (HIP events; host_alloc_and_wait: wall clock): two passes of "
                          "k_mc staging a 9^3 tile per block (count, emit), rocPRIM radix sorts of the 3T edge keys and the T cube keys, weld, download of %.0f MB through two page-locked "
                          "bounce buffers with a copy team" % (mc["blocks"], out_b / 1e6))
        r1 = first_run[0]
        return {"marching_cubes": mc, "frames": int(r1["frames_total"]), "frames_per_s": round(r1["frames_total"] / r1["seconds_total"], 1), "seconds": round(r1["seconds_total"], 4),
                "frames_per_s_best": round(rs["frames_total"] / rs["seconds_total"], 1), "writer": writer, "other_writer": other, "alternatives": alternatives,
                "compressed_bytes_per_frame": round(size / n), "jpeg_bytes_per_picture": jpeg_bytes, "sens_bytes": size, "decode_threads": int(rs["decode_threads"]),
                "decode_ms_per_frame_per_thread": round(1e3 * rs["seconds_decode_cpu"] / max(rs["frames_total"], 1), 3),
                "colour_fused": int(rs["color_fused"]), "frames_per_s_first_and_second_run": runs,
                "depth_inflated_on_device": int(rs.get("depth_inflated_on_device", -1)), "depth_inflated_on_host": int(rs.get("depth_inflated_on_host", -1)),
                "jpeg_entropy_on_device": int(rs.get("jpeg_entropy_on_device", -1)), "jpeg_entropy_on_host": int(rs.get("jpeg_entropy_on_host", -1)),
                "depth_inflate": "gpu (csrc/inflate_gpu.hip)", "inflate_kernels": kernels, "host_inflate": host_inflate,
                "blocks_live_end": st["blocks_allocated"], "alloc_failures": st["alloc_failures"], "write_s": round(t_write, 2),
                "page_cache": "warm: the file was written by this process seconds before the timed read and is mmapped (no disk I/O is timed; the stage behind "
                              "`convert` finds its input the same way)",
                "what": ".sens on disk written by: %s (zlib depth%s, %d KB per frame) -> %d host threads copy the compressed depth frames%s into the pinned ring -> H2D -> inflate on the GPU -> "
                        "pre-pass / allocation / compaction / integrate, 32 frames per pass; wall time of sf_fuse_run (first byte read -> last kernel complete); frames_per_s = the FIRST run "
                        "of the file in this process (it creates the run's streams and its pinned pool), frames_per_s_best = the better of two; other_writer: the same frames as this "
                        "library's writer compresses them; host_inflate: the same file with the host threads inflating"
                        % (writer, " + baseline-JPEG colour at 1296x968 with its own intrinsics, %d KB per picture" % (jpeg_bytes // 1024) if colour == "jpeg1296" else "", size // n // 1024,
                           rs["decode_threads"], " and strip the byte stuffing of the colour pictures' entropy-coded segments (Huffman decoding, IDCT, upsampling and RGB on the GPU; `alternatives`: the same file "
                           "with the entropy decoding on the host threads)" if colour == "jpeg1296" else "")}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def repeats_for(K, cfg_name="4mm"):
    """How often the K timed steps are repeated (fresh volume each time) so that the timed regions add up to ~1 s of GPU time at the rate this
    path runs at (~30 us per frame at 4 mm, ~1 ms at 1 mm) -- a function of K and the configuration alone, so every rank of a multi-GPU run takes
    the same number."""
    per_frame = 30e-6 if cfg_name == "4mm" else 1e-3
    return int(min(2000, max(3, -(-1.0 // (K * per_frame)))))


def run_stream(args, cfg_name, rank, local_rank, world, dist, torch):
    from scannet_amd import _abi, fusion, synth
    cfg = CONFIGS[cfg_name]
    K = args.steps if args.steps is not None else cfg["steps"]
    Wm = args.warmup if args.warmup is not None else cfg["warmup"]
    child = args.child
    # the roofline sample: the timed region itself when it is long enough to hold full 16-frame passes, else a fixed window of the same stream
    # (frames 64..463) -- a 20-step run times two short launches, which says nothing about the kernel
    if K >= 400 or child or cfg_name != "4mm":
        roof_W, roof_K = Wm, min(K, args.pmc_steps)
    else:
        roof_W, roof_K = 64, 400
    rgbd = not args.depth_only   # the metric's own configuration: a colour frame per depth frame
    e2e_n = 0 if (child or world > 1 or args.no_e2e or cfg_name != "4mm") else args.e2e_frames
    cpu_n = 0 if (child or world > 1 or args.no_cpu_baseline) else args.cpu_frames
    ooc_n = 0 if (child or world > 1 or args.no_out_of_cache or cfg_name != "4mm" or args.no_profile) else 16 + 64
    n_frames = max(K + Wm, roof_W + roof_K if not child else 0, e2e_n, cpu_n, ooc_n)
    first = (rank * 697) % TOTAL_FRAMES   # every rank walks the same room from a different starting frame (an independent scan per GPU)
    stride = W * H * 2
    frames = torch.empty((n_frames, H, W), dtype=torch.int16, device="cuda")
    poses = synth.render_scan_device(frames.data_ptr(), stride, first, n_frames, TOTAL_FRAMES, W, H, noise=args.noise, scene=args.scene, seed=0)
    params = fusion.default_params(voxel_size=cfg["voxel_size"], hash_num_buckets=cfg["hash_num_buckets"], num_sdf_blocks=cfg["num_sdf_blocks"])

    colour_frames = [None]

    def colour_tensor(n):
        """A synthetic RGB frame per depth frame, resident in HBM (uint8 [n, H, W, 3], 921 600 B per frame): gradients that move with the frame
        index under a per-pixel texture, a black band at the top (colour_first tells black from unobserved) and a saturated patch."""
        if colour_frames[0] is None or colour_frames[0].shape[0] < n:
            yy = torch.arange(H, device="cuda", dtype=torch.int32).view(1, H, 1)
            xx = torch.arange(W, device="cuda", dtype=torch.int32).view(1, 1, W)
            tex = ((xx * 7919 + yy * 104729) >> 3) & 31
            out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
            for a in range(0, n, 256):   # chunks: the int32 temporaries of 5 578 frames at once would be tens of GB
                b = min(n, a + 256)
                k = torch.arange(a, b, device="cuda", dtype=torch.int32).view(-1, 1, 1) + first
                out[a:b, ..., 0] = ((xx * 255 // W + 5 * k + tex) % 256).to(torch.uint8)
                out[a:b, ..., 1] = ((yy * 255 // H + 3 * k + tex) % 256).to(torch.uint8)
                out[a:b, ..., 2] = ((xx + yy + 7 * k + tex) % 256).to(torch.uint8)
            out[:, : H // 8] = 0
            out[:, H // 2: H // 2 + H // 16, : W // 3] = 255
            colour_frames[0] = out
        return colour_frames[0]

    def run(n_warm, n_timed, profile, single_frame=False, colour=None, extra_tune=None, repeats=1, prm=None):
        """`repeats` times on one fuser (sf_fuser_reset in between): fuse frames [0, n_warm) untimed, then frames [n_warm, n_warm + n_timed) between
        two barrier+synchronize pairs.  Elapsed times per repeat (max over ranks), kernel events and counters summed over the repeats."""
        colour = rgbd if colour is None else colour
        fuser = fusion.Fuser(prm if prm is not None else params, device=local_rank, **dict(dict(TUNE, **({"batch": 1} if single_frame else {})), **(extra_tune or {})))
        rgb = colour_tensor(n_warm + n_timed) if colour else None
        cstride = W * H * 3

        def sync_all():
            fuser.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()

        times, own_times, t_enq_sum, kernel_ms, launches, blocks, tiles = [], [], 0.0, 0.0, 0, 0, 0
        st1 = None
        for rep in range(repeats):
            if rep:
                fuser.reset()
            fuser.integrate_batch_device(frames[:n_warm].data_ptr(), stride, poses[:n_warm], rgb.data_ptr() if colour else None, cstride)
            sync_all()
            st0 = fuser.stats()
            if profile:
                fuser.profile(True)
            sync_all()
            t0 = time.perf_counter()
            fuser.integrate_batch_device(frames[n_warm:].data_ptr(), stride, poses[n_warm:n_warm + n_timed], rgb[n_warm:].data_ptr() if colour else None, cstride)
            t_enq_sum += time.perf_counter() - t0
            sync_all()
            elapsed = time.perf_counter() - t0
            own_times.append(elapsed)
            if world > 1:
                tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elapsed = float(tt.item())
            times.append(elapsed)
            st1 = fuser.stats()
            if profile:
                ms, ln, _ = fuser.profile_read()
                kernel_ms += ms
                launches += ln
                fuser.profile(False)
            blocks += st1["total_frame_blocks"] - st0["total_frame_blocks"]
            tiles += st1["total_pass_tiles"] - st0["total_pass_tiles"]
        ceiling = None
        if single_frame and profile:
            # the last frame's tile traffic without the arithmetic: what this access pattern (scattered 4 KiB RMW) can reach
            rmw_us, ntiles = fuser.calib_tile_rmw(read_only=False, iters=20)
            ro_us, _ = fuser.calib_tile_rmw(read_only=True, iters=20)
            if ntiles > 0:
                ceiling = {"tiles": ntiles, "rmw_copy_us": round(rmw_us, 2), "rmw_copy_GBs": round(ntiles * 8192 / rmw_us / 1e3, 1),
                           "read_only_us": round(ro_us, 2), "read_only_GBs": round(ntiles * 4096 / ro_us / 1e3, 1)}
                if ntiles * 4096 > (512 << 20) and hasattr(fuser, "calib_tile_rmw_ex"):   # out of cache: what holds the read-modify-write below the read-only rate?
                    dec = {}
                    for label, kw in (("scattered_rmw", {}), ("scattered_rmw_2_tiles_per_turnaround", {"tiles_per_turnaround": 2}), ("scattered_rmw_4_tiles_per_turnaround", {"tiles_per_turnaround": 4}),
                                      ("contiguous_rmw", {"contiguous": True}), ("contiguous_rmw_4_tiles_per_turnaround", {"contiguous": True, "tiles_per_turnaround": 4}),
                                      ("scattered_read_only", {"read_only": True}), ("contiguous_read_only", {"read_only": True, "contiguous": True})):
                        try:
                            us, nt_ = fuser.calib_tile_rmw_ex(iters=8, **kw)
                            dec[label] = {"us": round(us, 1), "GBs": round(nt_ * (4096 if kw.get("read_only") else 8192) / us / 1e3, 1)}
                        except Exception as ex:
                            dec[label] = {"error": str(ex)[:120]}
                    ceiling["rmw_decomposition"] = dec
                    ceiling["rmw_decomposition_what"] = ("the pass's tile traffic without arithmetic, taken apart: the pass's scattered 4 KiB tiles or one contiguous span of as many tiles "
                                                         "(tiles 0..n-1 of the pool); every tile written back right after it is read, or 2 / 4 tiles read and then written (fewer read/write "
                                                         "turnarounds in flight per wave); read only.  Every tile is written back as read")
        batch = fuser.batch_frames
        fuser.close()
        # B_frame = N_blk*(512*8 read + 512*8 write + 16) + W*H*2 + 64, summed over the frames (and the repeats)
        alg_bytes = blocks * (4096 + 4096 + 16) + repeats * n_timed * (W * H * (5 if colour else 2) + 64)
        # what ONE pass over the tiles has to move whatever the number of frames it fuses: every tile of the pass's list read and written once
        # plus the frames' depth images -- the memory roofline of the temporally blocked launch
        batch_bytes = tiles * 8192 + repeats * n_timed * (W * H * (5 if colour else 2))
        tsort = sorted(times)
        return {"elapsed": tsort[len(tsort) // 2], "times": times, "own_times": own_times, "t_enq": t_enq_sum / repeats, "kernel_ms": kernel_ms, "launches": launches, "blocks": blocks,
                "alg_bytes": alg_bytes, "batch_bytes": batch_bytes, "tiles": tiles, "repeats": repeats,
                "batch": batch, "n_launch": (n_timed + batch - 1) // batch, "st1": st1, "ceiling": ceiling, "colour": colour}

    def per_launch(m, n_timed):
        return {"avg_kernel_us": round(m["kernel_ms"] * 1e3 / m["launches"], 2), "launches": m["launches"],
                "frames_per_launch": round(m["repeats"] * n_timed / m["launches"], 2), "avg_frame_blocks_per_launch": round(m["blocks"] / m["launches"], 1),
                "alg_bytes_per_launch": round(m["alg_bytes"] / m["launches"])}

    def roofline_hbm(m, n_timed, kernel):
        """One frame per launch: HBM bound, achieved = SURVEY 8d algorithmic bytes over the launch duration (HIP events on the fuser's stream)."""
        if not m["launches"]:
            return None
        achieved = m["alg_bytes"] / (m["kernel_ms"] * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
             "traffic": None, "kernel": kernel}
        r.update(per_launch(m, n_timed))
        fp = m["blocks"] / m["launches"] * 4096
        r["tile_footprint_bytes_per_launch"] = round(fp)
        r["footprint_vs_infinity_cache"] = round(fp / MALL_BYTES, 2)
        r["footprint_note"] = ("the launch's tile set is %.1f x the 256 MiB Infinity Cache: %s" %
                               (fp / MALL_BYTES, "tiles come from HBM every launch" if fp > 4 * MALL_BYTES else
                                "consecutive frames re-touch tiles that may still be on-die (FETCH_SIZE counts those hits): see roofline_out_of_cache"))
        return r

    def roofline_valu(m, n_timed, kernel, valu, traffic, sample):
        """Many frames per launch (32 by default): VALU-issue bound.  frac = measured VALU issue utilisation; hbm_frac from the counter traffic; hbm_alg_batch = the
        bytes one pass must move (its tiles once + its depth images) over the launch duration, against the 8 TB/s peak."""
        if not m["launches"]:
            return None
        t_s = m["kernel_ms"] * 1e-3 / m["launches"]
        alg_equiv = m["alg_bytes"] / m["launches"] / t_s / 1e9
        batch_gbs = m["batch_bytes"] / m["launches"] / t_s / 1e9
        # frac: the VALU time the launch's instructions need at the issue costs MEASURED on this part (valu_peak_calibration) over the SIMD time of the launch.
        # Rounds 2-4 reported SQ_ACTIVE_INST_VALU x 4 / SIMD cycles ("a wave64 instruction holds its SIMD for 4 cycles"); the microbenchmark shows that
        # counter ticks once per instruction and that the simple fp32 / integer instructions take ~2.2 cycles: that ratio (kept as issue_ratio_4_cycles)
        # reaches 1.8 on a pure v_fma stream and is not a utilisation.
        cal = (valu or {}).get("calibrated")
        # frac: SQ_INSTS_VALU x 2 cycles (the guide's wave64 issue rate, MI355X_MICROARCH.md, CU section) over the SIMD cycles of the
        # launch -- a fraction of the guide's peak, <= 1 by construction.  The class-priced figure of round 5 (every instruction at its measured issue cost)
        # stays beside it as frac_class_priced: it prices the launch at ~100 % of its time and is a model of where the cycles go, not a ceiling.
        frac = valu.get("valu_frac_2cycle") if valu else None
        bound, unit, achieved, peak = "valu", "SQ_INSTS_VALU x 2 cycles / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)", frac, 1.0
        if frac is None:   # no counter pass (no rocprofv3, --no-pmc, N > 1): the memory roofline of the pass, from its must-move bytes and the HIP-event duration
            bound, unit, achieved, peak, frac = "hbm", "GB/s", round(batch_gbs, 1), HBM_PEAK_GBS, round(batch_gbs / HBM_PEAK_GBS, 4)
        r = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": frac,
             "valu_frac_2cycle": valu.get("valu_frac_2cycle") if valu else None, "frac_class_priced": cal["frac_serial"] if cal else None,
             "valu_peak_calibration": cal, "issue_ratio_4_cycles": valu["valu_util"] if valu else None,
             "traffic": traffic["bytes"] if traffic else None, "kernel": kernel, "sample": sample,
             "hbm_frac": round(traffic["bytes"] / t_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
             "hbm_GBs": round(traffic["bytes"] / t_s / 1e9, 1) if traffic else None,
             "hbm_alg_batch": {"bytes_per_launch": round(m["batch_bytes"] / m["launches"]), "tiles_per_launch": round(m["tiles"] / m["launches"], 1),
                               "GBs": round(batch_gbs, 1), "frac": round(batch_gbs / HBM_PEAK_GBS, 4),
                               "what": "(tiles of the pass x 8192 B + frames of the pass x %d B) / launch duration / 8 TB/s: the memory roofline of the temporally "
                                       "blocked pass -- comparable across batch sizes, <= 1" % (W * H * 2)},
             "alg_equiv_GBs": round(alg_equiv, 1),
             "alg_equiv_note": "SURVEY 8d algorithmic bytes of frame-by-frame fusion over the launch duration: what the launch replaces, not bytes it moves "
                               "(temporal blocking keeps each tile in registers for all frames of the batch) -- not a roofline fraction"}
        r.update(per_launch(m, n_timed))
        if valu:
            r["valu_detail"] = valu
        if traffic:
            r["traffic_detail"] = traffic
        # the kernels in front of the integrate launch (allocation, compaction, pre-pass; one launch each per pass): VALU and HBM figures from the
        # same counter passes -- each measured ALONE (the profiler serialises kernels); in the shipped schedule they run beside the previous integrate
        fc = {}
        for src in (valu, traffic):
            for k, e in ((src or {}).get("front_chain") or {}).items():
                fc.setdefault(k, {}).update(e)
        if fc:
            r["front_chain"] = fc
        return r

    R = 1 if child else (args.repeats if args.repeats else repeats_for(K, cfg_name))
    xr = "true" if TUNE.get("xrow", 1) else "false"
    kname = ("k_integrate<1,2,true,2,false,4,%s>" if rgbd else "k_integrate<1,0,true,2,false,4,%s>") % xr
    # the timed region runs WITHOUT the HIP events around every integrate launch (they cost ~2 % of the frames/s: profiles/r03_small_experiments.txt);
    # kernel durations come from the roofline sample below, fused again with the events on.  --single-frame keeps them: its line is the roofline.
    m = run(Wm, K, args.single_frame and not args.no_profile, single_frame=args.single_frame, repeats=R)
    # the roofline window, on EVERY rank (run() holds barriers and an all-reduce when N > 1: a pass only rank 0 entered would hang the job)
    mr = m if (child or args.no_profile or m["batch"] == 1) else run(roof_W, roof_K, True)
    # every rank's own rate, gathered (N > 1): the driver computes scaling efficiency from `value`, a reader sees the spread here
    per_rank = None
    if world > 1:
        mine = torch.tensor([K / min(m["own_times"])], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [round(float(t.item()), 1) for t in allr]
    out = None
    if rank == 0:
        pmc_on = world == 1 and not args.no_pmc and not args.no_profile and not child
        if m["batch"] == 1:
            roof = roofline_hbm(m, K, "k_integrate_pipe<true,2,*>" if not rgbd else "k_integrate<1,2,true,2,true> (one frame per launch)")
            if roof is not None and m["ceiling"]:
                roof["pattern_ceiling"] = dict(m["ceiling"], frac_of_ceiling=round(roof["achieved"] / m["ceiling"]["rmw_copy_GBs"], 4))
        else:
            same = (roof_W, roof_K) == (Wm, K)
            valu = pmc_valu(args, cfg_name, roof_K, roof_W, False, not rgbd) if pmc_on else None
            traffic = pmc_traffic(args, cfg_name, roof_K, roof_W, False, not rgbd) if pmc_on else None
            mempipe = pmc_mempipe(args, cfg_name, roof_K, roof_W, False, not rgbd) if pmc_on else None
            roof = roofline_valu(mr, roof_K, kname, valu, traffic,
                                 "frames %d..%d of the stream (%s), HIP events around every integrate launch; counters from rocprofv3 --pmc passes over the same frames"
                                 % (roof_W, roof_W + roof_K - 1, "the timed region, fused again with events on" if same else
                                    ("the first frames of the timed region, fused again with events on" if roof_W == Wm else
                                     "a fixed window: the timed region of this run is too short to hold full passes")))
            if roof is not None:
                roof["mem_pipe"] = mempipe
                roof["ta_busy"] = mempipe["ta_busy"] if mempipe else None
                roof["tcp_tag_lookups_per_cu_clk"] = mempipe["tcp_tag_lookups_per_cu_clk"] if mempipe else None
                must = roof["hbm_alg_batch"]["bytes_per_launch"]
                roof["traffic_over_must_move"] = round(traffic["bytes"] / must, 4) if (traffic and must) else None
        ts = m["times"]
        px_bytes = 5 if rgbd else 2
        out = {
            "metric": "RGB-D frames/sec integrated (640x480, %s voxel)" % ("4 mm" if cfg_name == "4mm" else "1 mm"),
            "value": round(world * K / m["elapsed"], 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(m["elapsed"] * 1e3 / max(K, 1), 5),
            "host_enqueue_ms_per_step": round(m["t_enq"] * 1e3 / max(K, 1), 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else (0 if world > 1 else 1),
            "process_group": (dist.get_backend() if world > 1 else "none (one process)"),
            "per_rank_frames_per_s": per_rank,
            "repeats": {"n": len(ts), "timed_s_total": round(sum(ts), 4), "value_median": round(world * K / m["elapsed"], 2),
                        "value_min": round(world * K / max(ts), 2), "value_max": round(world * K / min(ts), 2),
                        "what": "the %d warm-up + %d timed steps repeated on an emptied volume (sf_fuser_reset) until the timed regions add up to ~1 s; "
                                "`value` is the median repeat" % (Wm, K)},
            "config": {"workload": cfg["label"] + ", frames %d..%d per rank, %s resident in HBM" % (Wm, K + Wm - 1, "depth AND colour (a 640x480 RGB8 frame per depth frame, 921 600 B)" if rgbd else "depth (geometry only: --depth-only)"),
                       "rgbd": rgbd, "tune": dict(TUNE),
                       "scene": {0: "empty box room", 1: "box room furnished with 48 boxes (csrc/synth.hip clutter_boxes), sensor holes"}[args.scene],
                       "noise": {0: "none", 1: "round-1 LCG ramp (3 LSBs)", 2: "3 LSBs hashed per pixel and frame"}[args.noise],
                       "sharding": "one independent scan per GPU, no collective on the data path",
                       "blocks_live_end": m["st1"]["blocks_allocated"], "alloc_failures": m["st1"]["alloc_failures"],
                       "frames_per_pass": m["batch"], "integrate_launches": m["n_launch"],
                       "alg_bytes_per_launch": round(m["alg_bytes"] / max(m["n_launch"] * m["repeats"], 1))},
            "roofline_inputs": {"block_frames_per_launch": round(m["blocks"] / max(m["n_launch"] * m["repeats"], 1), 1), "input_bytes_per_pixel": px_bytes},
            "roofline": roof,
        }
        if rgbd:
            out["value_rgbd"] = out["value"]
        else:
            out["value_depth_only"] = out["value"]
        extras = world == 1 and not args.no_profile and not args.single_frame and not child
        if extras and rgbd and not args.no_depth_only and cfg_name == "4mm":
            # the geometry-only pass over the same frames (the headline of rounds 1-3): rate with the same repeats, kernel duration and VALU counters
            md = run(Wm, K, False, colour=False, repeats=R)
            out["value_depth_only"] = round(K / md["elapsed"], 2)
            mdr = run(roof_W, roof_K, True, colour=False)
            vd = pmc_valu(args, cfg_name, roof_K, roof_W, False, True) if pmc_on else None
            rd = roofline_valu(mdr, roof_K, "k_integrate<1,0,true,2,false,4,%s>" % xr, vd, None,
                               "frames %d..%d of the stream without the colour frames, HIP events around every integrate launch" % (roof_W, roof_W + roof_K - 1))
            if rd is not None:
                rd["frames_per_s"] = out["value_depth_only"]
                rd["colour_pass_over_depth_pass"] = round(mr["kernel_ms"] / mr["launches"] / (mdr["kernel_ms"] / mdr["launches"]), 4) if (mr["launches"] and mdr["launches"]) else None
                out["roofline_depth_only"] = rd
        if extras and not args.no_single_frame:
            # the same update HBM-bound: one frame per launch (what sf_fuser_integrate does for a live stream), on the roofline sample; geometry only --
            # the colourless one-frame pass runs the software-pipelined kernel, whose bytes are SURVEY 8d's formula
            ks = min(roof_K, 1200)
            m1 = run(roof_W, ks, True, single_frame=True, colour=False)
            r1 = roofline_hbm(m1, ks, "k_integrate_pipe<true,2,*>: one DEPTH frame per launch (batch = 1), persistent, software-pipelined (tiles and depth gathers "
                                      "of later tiles in flight into LDS)")
            if r1 is not None:
                r1["frames_per_s"] = round(ks / m1["elapsed"], 1)
                r1["ms_per_frame"] = round(m1["elapsed"] * 1e3 / ks, 5)
                r1["sample"] = "frames %d..%d" % (roof_W, roof_W + ks - 1)
                if m1["ceiling"]:
                    r1["pattern_ceiling"] = dict(m1["ceiling"], frac_of_ceiling=round(r1["achieved"] / m1["ceiling"]["rmw_copy_GBs"], 4),
                                                 note="k_tile_rmw: the same tiles of the last timed frame read and written back unchanged, no arithmetic, same "
                                                      "launch geometry -- what scattered 4 KiB read-modify-write reaches on this HBM")
                if pmc_on:
                    t = pmc_traffic(args, cfg_name, min(roof_K, ks), roof_W, True, True)
                    if t is not None:
                        r1["traffic"] = t["bytes"]
                        r1["traffic_detail"] = t
                if rgbd:
                    m1c = run(roof_W, min(ks, 400), True, single_frame=True, colour=True)
                    if m1c["launches"]:
                        r1["rgbd_one_frame_per_launch"] = {"frames_per_s": round(min(ks, 400) / m1c["elapsed"], 1), "avg_kernel_us": round(m1c["kernel_ms"] * 1e3 / m1c["launches"], 2),
                                                           "kernel": "k_integrate<1,2,true,2,true>",
                                                           "hbm_frac_alg": round(m1c["alg_bytes"] / (m1c["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                # the entry point a live stream calls: one pageable host frame per call (sf_fuser_integrate), through the page-locked ring
                nh = min(ks, 400)
                host = frames[roof_W:roof_W + nh].cpu().numpy().view(np.uint16)
                with fusion.Fuser(params, device=local_rank, **TUNE) as fh:
                    fh.integrate_batch_device(frames[:roof_W].data_ptr(), stride, poses[:roof_W])
                    fh.sync()
                    t0 = time.perf_counter()
                    for i in range(nh):
                        fh.integrate(host[i], poses[roof_W + i])
                    fh.sync()
                    dt = time.perf_counter() - t0
                r1["live_stream_host_buffers"] = {"frames_per_s": round(nh / dt, 1), "frames": nh,
                                                  "what": "sf_fuser_integrate per depth frame from pageable host memory (copy into a page-locked ring slot, H2D, "
                                                          "pre-pass, allocation, compaction, integrate queued; no stream drained per frame), PCIe included"}
                out["roofline_single_frame"] = r1
        if e2e_n:
            # the metric is RGB-D: the leg named end_to_end is the WHOLE scan with ScanNet's real colour stream (1296x968 baseline JPEG over zlib depth), the first
            # sf_fuse_run of this process (the 5 578-frame leg was depth only and the RGB-D one 1 024 frames); the geometry-only file of the same
            # frames (the same reference-written depth streams) follows as end_to_end.depth_only
            zcache = {}
            e_rgbd = None
            if rgbd and not args.no_e2e_rgbd:
                n_rgbd = min(e2e_n, args.e2e_rgbd_frames)
                if shutil.disk_usage("/tmp").free < 2 * n_rgbd * 700000:   # ~590 KB per frame, and the depth-only files behind it
                    n_rgbd = min(n_rgbd, 1024)
                try:
                    e_rgbd = end_to_end(frames, poses, n_rgbd, params, local_rank, torch, colour="jpeg1296", cache=zcache)
                except Exception as ex:   # a leg beside the metric (disk full in /tmp, ...): never take the line down
                    e_rgbd = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            try:
                e_depth = end_to_end(frames, poses, e2e_n, params, local_rank, torch, cache=zcache)
            except Exception as ex:
                e_depth = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            zcache.clear()
            if e_rgbd is not None:
                out["end_to_end"] = dict(e_rgbd, depth_only=e_depth)
                out["end_to_end_rgbd"] = {k: e_rgbd.get(k) for k in ("frames_per_s", "frames_per_s_best", "frames", "colour_fused", "decode_threads", "error") if k in e_rgbd}
                out["end_to_end_rgbd"]["what"] = "the same leg as end_to_end (kept under the name earlier rounds reported it by)"
            else:
                out["end_to_end"] = e_depth
        if extras and ooc_n:
            # (behind the end-to-end legs: this leg's fusers open the second front stream at the device's lowest priority, and the first stream of a priority class makes the
            # runtime open that class's hardware queues for the life of the process -- sf_fuse_run's seven to nine busy streams then run ~12 % slower)
            # BASELINE configs[2] bounded: the same first frames at 1 mm voxels (2^22 buckets, 2^25 blocks = 137 GB of tiles reserved), one frame per
            # launch -- every launch's tile set is ~20 x the Infinity Cache, so this IS an HBM figure
            c1 = CONFIGS["1mm"]
            p1 = fusion.default_params(voxel_size=c1["voxel_size"], hash_num_buckets=c1["hash_num_buckets"], num_sdf_blocks=c1["num_sdf_blocks"])
            try:
                mo = run(16, 64, True, single_frame=True, prm=p1, colour=False)
                ro = roofline_hbm(mo, 64, "k_integrate_pipe<true,2,nt>: one depth frame per launch at 1 mm voxels, non-temporal tile traffic")
                if ro is not None:
                    ro["frames_per_s"] = round(64 / mo["elapsed"], 1)
                    ro["sample"] = "frames 16..79 of the same stream at 1 mm voxels (BASELINE configs[2] geometry); `python bench.py --config 1mm` is the long form"
                    if mo["ceiling"]:
                        ro["pattern_ceiling"] = dict(mo["ceiling"], frac_of_ceiling=round(ro["achieved"] / mo["ceiling"]["rmw_copy_GBs"], 4))
                    ma = run(16, 64, True, single_frame=True, prm=p1, extra_tune={"pipe_overlap": 0}, colour=False)
                    if ma["launches"]:
                        ach = ma["alg_bytes"] / (ma["kernel_ms"] * 1e-3) / 1e9
                        ro["kernel_alone"] = {"tune": "pipe_overlap=0 (the next frame's pre-pass / allocation / compaction serialised behind the kernel)",
                                              "avg_kernel_us": round(ma["kernel_ms"] * 1e3 / ma["launches"], 2), "achieved": round(ach, 1),
                                              "frac": round(ach / HBM_PEAK_GBS, 4), "frames_per_s": round(64 / ma["elapsed"], 1),
                                              "frac_of_ceiling": round(ach / mo["ceiling"]["rmw_copy_GBs"], 4) if mo["ceiling"] else None}
                    # the 32-frame schedule on the same frames, for the frames/s of configs[2]
                    mb = run(16, 64, True, prm=p1, colour=False)
                    ro["batched_frames_per_s"] = round(64 / mb["elapsed"], 1)
                    out["roofline_out_of_cache"] = ro
                    # the numbers themselves inside `roofline` (the driver's record keeps that object, of the extra ones only the key names)
                    if roof is not None:
                        roof["hbm_out_of_cache"] = {"frac": ro["frac"], "achieved_GBs": ro["achieved"], "frac_of_rmw_ceiling": (ro.get("pattern_ceiling") or {}).get("frac_of_ceiling"),
                                                    "kernel_alone_frac": (ro.get("kernel_alone") or {}).get("frac"), "footprint_vs_infinity_cache": ro["footprint_vs_infinity_cache"],
                                                    "rmw_decomposition": (ro.get("pattern_ceiling") or {}).get("rmw_decomposition"),
                                                    "what": "k_integrate_pipe, one depth frame per launch at 1 mm voxels (BASELINE configs[2] geometry): SURVEY 8d algorithmic bytes / launch "
                                                            "duration / 8 TB/s in the shipped schedule; details under roofline_out_of_cache"}
            except _abi.ScanfuseError as e:   # a smaller GPU than the 288 GB part cannot reserve the tiles
                out["roofline_out_of_cache"] = {"error": str(e)}
        if cpu_n:   # rank 0 at N = 1 only
            host = frames[:cpu_n].cpu().numpy().view(np.uint16)
            rgb_dev = colour_tensor(max(cpu_n, 1)) if rgbd else None
            rgb_host = rgb_dev[:cpu_n].cpu().numpy() if rgbd else None
            base, ovol, n_cpu = cpu_baseline(host, poses[:cpu_n].reshape(-1, 4, 4), cfg["voxel_size"], frames=cpu_n, rgb_host=rgb_host)
            out["cpu_baseline"] = base
            out["parity"] = parity_leg(ovol, n_cpu, frames, stride, poses, params, local_rank, rgb_dev=rgb_dev)
            ovol.close()
            if rgbd:
                out["parity_colour"] = out["parity"]
                # and the geometry-only path of the same build against the oracle's geometry-only volume (all threads, not timed)
                from oracle import oracle as orc
                vol = orc.Volume(orc.default_params(W, H, cfg["voxel_size"]), threads=_abi.usable_cpus())
                for i in range(n_cpu):
                    vol.integrate(host[i], poses[i].reshape(4, 4))
                out["parity_depth_only"] = parity_leg(vol, n_cpu, frames, stride, poses, params, local_rank)
                vol.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# configs[3]: independent scans, scan-per-GPU, one shared longest-first queue, no collective
# ---------------------------------------------------------------------------------------------------------------------------------------
def run_scans(args, rank, local_rank, world, dist, torch):
    from scannet_amd import _abi, fusion, meshclean, segmentator, shard, synth
    K = args.steps if args.steps is not None else 12      # scans per GPU (weak scaling: N x K scans in the queue)
    Wm = args.warmup if args.warmup is not None else 1
    n_scans = K * world
    specs = [synth.scan_spec(args.first_scan + i) for i in range(n_scans)]
    if args.max_scan_frames:
        specs = [(room, min(n, args.max_scan_frames)) for room, n in specs]
    costs = [n for _, n in specs]
    stride = W * H * 2
    chunk = 2048
    buf = torch.empty((chunk, H, W), dtype=torch.int16, device="cuda")
    params = fusion.default_params()
    gpu_busy = [0.0]
    render_busy = [0.0]   # of which: rendering the synthetic input (sf_synth_scene_device synchronises) -- benchmark scaffolding, not the path
    frames_done = [0]
    outdir = tempfile.mkdtemp(prefix="sf_scans_r%d_" % rank, dir="/tmp")

    def gpu_stage(i):
        room, n = specs[i]
        t0 = time.perf_counter()
        with fusion.Fuser(params, device=local_rank, **TUNE) as f:
            for a in range(0, n, chunk):
                m = min(chunk, n - a)
                tr = time.perf_counter()
                poses = synth.render_scan_device(buf.data_ptr(), stride, a, m, n, W, H, room=room, noise=args.noise, scene=args.scene, seed=args.first_scan + i)
                render_busy[0] += time.perf_counter() - tr
                f.integrate_batch_device(buf.data_ptr(), stride, poses)
                f.sync()
            mesh = f.extract_mesh()
        gpu_busy[0] += time.perf_counter() - t0
        frames_done[0] += n
        return mesh

    def host_stage(i, mesh):
        t0 = time.perf_counter()
        nv, nf = mesh.counts()
        res = {"scan": args.first_scan + i, "frames": specs[i][1], "faces": nf}
        if args.host_stage != "none":
            ta = time.perf_counter()
            gpu_filters = local_rank if args.host_stage == "gpu" else None
            cleaned, _ = meshclean.clean(mesh, meshclean.CLEAN_MLX_MERGE_DISTANCE, 7500, gpu=gpu_filters)
            res["clean_s"] = time.perf_counter() - ta
            cur = cleaned
            res["decimate_s"] = res["clean_lores_s"] = 0.0
            if args.host_stage in ("full", "gpu-decimate", "gpu"):
                for _ in range(2):
                    ta = time.perf_counter()
                    simp, sst = meshclean.simplify(cur, gpu=local_rank if args.host_stage in ("gpu-decimate", "gpu") else None)
                    tb = time.perf_counter()
                    cur, _ = meshclean.clean(simp, meshclean.CLEAN_MLX_MERGE_DISTANCE, meshclean.CLEAN_LORES_MIN_COMPONENT, gpu=gpu_filters)
                    res["decimate_s"] += tb - ta
                    res["clean_lores_s"] += time.perf_counter() - tb
                    res["decimate_rounds"] = res.get("decimate_rounds", 0) + sst["rounds"]
            ta = time.perf_counter()
            ply = os.path.join(outdir, "scan%04d_vh_clean_2.ply" % i)
            cur.write_ply(ply)
            res["segments"] = segmentator.segment_to_json(ply, 0.01, 20)
            res["segment_s"] = time.perf_counter() - ta
            res["faces_out"] = cur.counts()[1]
        res["host_s"] = time.perf_counter() - t0
        return res

    workers = max(1, _abi.usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))
    for w in range(Wm):   # untimed: pages in the library, the kernels and the host stage
        host_stage(0, gpu_stage(0)) if args.host_stage != "none" else gpu_stage(0)
    gpu_busy[0], frames_done[0], render_busy[0] = 0.0, 0, 0.0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    done = shard.run_pipelined(list(range(n_scans)), costs, gpu_stage, host_stage, workers, key="scanfuse/bench/scans")
    t_gpu_done = time.perf_counter() - t0   # run_pipelined returns when the host pool has drained too
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stats = torch.tensor([elapsed, gpu_busy[0], float(frames_done[0]), float(len(done))], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, busy_sum, frames_sum, scans_sum = float(mx[0]), float(sm[1]), float(sm[2]), float(sm[3])
    else:
        busy_sum, frames_sum, scans_sum = gpu_busy[0], float(frames_done[0]), float(len(done))
    shutil.rmtree(outdir, ignore_errors=True)
    if rank != 0:
        return None
    host_s = [r["host_s"] for _, r in done]
    return {
        "metric": "scans/min rebuilt (independent 640x480 scans, 4 mm voxel, scan-per-GPU)",
        "value": round(scans_sum / elapsed * 60.0, 3), "unit": "scans/min", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(elapsed * 1e3 / max(K, 1), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "frames_per_s": round(frames_sum / elapsed, 1),
        "config": {"workload": "configs[3]: scans %d..%d of the 1513-scan synthetic rebuild (room size +-20 %%, 300..6000 frames each, 640x480, 4 mm voxels), "
                               "%d per GPU popped longest-first from ONE queue in the rendezvous store" % (args.first_scan, args.first_scan + n_scans - 1, K),
                   "sharding": "scan-per-GPU, no collective on the data path",
                   "host_stage": {"full": "clean.mlx + quadric collapse to 20 %% twice + cleanLoRes + Segmentator per scan on a pool of %d host threads per rank" % workers,
                                  "gpu-decimate": "clean.mlx + quadric collapse to 20 %% twice ON THE GPU (rounds of independent collapses) + cleanLoRes + Segmentator, "
                                                  "driven by a pool of %d host threads per rank" % workers,
                                  "gpu": "clean.mlx, quadric collapse to 20 %% twice and cleanLoRes ALL ON THE GPU (sf_mesh_clean_gpu, sf_mesh_simplify_gpu) + Segmentator, "
                                         "driven by a pool of %d host threads per rank" % workers,
                                  "clean": "clean.mlx + Segmentator per scan on a pool of %d host threads per rank" % workers, "none": "none (fusion + marching cubes only)"}[args.host_stage],
                   "frames_total": int(frames_sum)},
        "gpu_busy_s_sum": round(busy_sum, 3), "gpu_idle_pct": round(100.0 * (1.0 - busy_sum / (elapsed * world)), 1),
        "input_render_s_rank0": round(render_busy[0], 3),
        "input_render_note": "seconds of rank 0's GPU-busy time spent RENDERING the synthetic scans (csrc/synth.hip: benchmark input, counted in the rate because it "
                             "occupies the same GPU; a real rebuild reads .sens files instead)",
        "host_stage_s_mean_rank0": round(float(np.mean(host_s)), 3) if host_s else None,
        "host_stage_parts_s_mean_rank0": {k: round(float(np.mean([r.get(k, 0.0) for _, r in done])), 3)
                                          for k in ("clean_s", "decimate_s", "clean_lores_s", "segment_s", "decimate_rounds", "faces")} if host_s else None,
        "roofline": None,
        "note": "per scan: frames rendered into HBM in chunks of %d (input generation, counted as GPU-busy), fused 32 frames per pass, marching cubes; the host "
                "stage runs on threads while the GPU takes the next scan.  With the full host stage the CPUs, not the GPU, set scans/min" % chunk,
    }


# ---------------------------------------------------------------------------------------------------------------------------------------
# configs[4]: one long scan, block space dealt in stripes to the ranks, boundary all-gather (RCCL) before marching cubes
# ---------------------------------------------------------------------------------------------------------------------------------------
def run_partition(args, rank, local_rank, world, dist, torch):
    from scannet_amd import fusion, partition, synth
    total = args.scan_frames
    Wm = args.warmup if args.warmup is not None else 64
    K = args.steps if args.steps is not None else total - Wm
    n_frames = K + Wm
    stride = W * H * 2
    frames = torch.empty((n_frames, H, W), dtype=torch.int16, device="cuda")   # every rank holds every frame (614 KB each)
    poses = np.zeros((n_frames, 16), np.float32)
    a = 0
    while a < n_frames:   # room by room
        room, inside, per = synth.corridor_room(a, total)
        m = min(per - inside, n_frames - a)
        poses[a:a + m] = synth.render_scan_device(frames[a:].data_ptr(), stride, inside, m, per, W, H, origin=(room * synth.CORRIDOR_PITCH, 0.0, 0.0),
                                                 noise=args.noise, scene=args.scene, seed=room)
        a += m
    blocks = 1 << 23 if world == 1 else max(1 << 20, (1 << 23) // world * 2)
    params = fusion.default_params(num_sdf_blocks=blocks, hash_num_buckets=max(1 << 19, blocks // 2))
    fuser = fusion.Fuser(params, device=local_rank, **TUNE)
    if world > 1 or args.stripes_at_one:
        if args.partition == "stripes":
            fuser.set_stripes(0, 0, args.stripe_blocks, max(world, 1), rank)
        else:
            planes = partition.slab_planes(0, int(np.ceil(synth.CORRIDOR_ROOMS * synth.CORRIDOR_PITCH / (8 * params.voxel_size))), world)
            fuser.set_slab(0, planes[rank], planes[rank + 1])

    def sync_all():
        fuser.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    fuser.integrate_batch_device(frames[:Wm].data_ptr(), stride, poses[:Wm])
    sync_all()
    t0 = time.perf_counter()
    fuser.integrate_batch_device(frames[Wm:].data_ptr(), stride, poses[Wm:])
    sync_all()
    t_fuse = time.perf_counter() - t0
    # the exchange step + meshing (reported beside the metric, not part of the K timed steps)
    t1 = time.perf_counter()
    sent, got = partition.exchange_boundary(fuser, mode=args.exchange) if world > 1 else (fuser.count_boundary(), 0)
    bytes_in = partition.exchange_boundary.last_bytes if world > 1 else 0
    sync_all()
    t_exch = time.perf_counter() - t1
    t2 = time.perf_counter()
    mesh = fuser.extract_mesh()
    nv, nf = mesh.counts()
    sync_all()
    t_mc = time.perf_counter() - t2
    st = fuser.stats()
    vals = torch.tensor([t_fuse, t_exch, t_mc, float(st["blocks_allocated"]), float(sent), float(got), float(nf), float(st["alloc_failures"]), float(bytes_in)],
                        dtype=torch.float64, device="cuda")
    if world > 1:
        mx = vals.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        mn = vals.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    else:
        mx = sm = mn = vals
    fuser.close()
    prefix = None
    if world > 1 and not args.no_prefix_check:
        prefix = partition_prefix_check(args, frames, stride, poses, params, rank, local_rank, world, dist)
    if rank != 0:
        return None
    t_fuse = float(mx[0])
    if world > 1 and args.exchange == "neighbour" and int(sm[4]) != int(sm[5]):
        raise SystemExit("bench.py: %d boundary blocks sent but %d ghost blocks received: the ring shift lost or duplicated layers" % (int(sm[4]), int(sm[5])))
    return {
        "metric": "RGB-D frames/sec integrated (640x480, 4 mm voxel), one scan partitioned over the GPUs",
        "value": round(K / t_fuse, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(t_fuse * 1e3 / K, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[4]: one %d-frame walk through a corridor of %d box rooms (6x4x3 m, %.1f m pitch), 640x480, 4 mm voxels; every rank sees every frame"
                               % (total, synth.CORRIDOR_ROOMS, synth.CORRIDOR_PITCH),
                   "partition": ("stripes of %d block layers along x dealt round-robin" % args.stripe_blocks) if args.partition == "stripes" else "one contiguous slab along x per rank",
                   "collective": ({"neighbour": "RCCL send / recv of the boundary block layers (device tensors) to the ONE rank that needs them (ring shift over the "
                                                "direct xGMI links), once, before marching cubes",
                                   "all_gather": "RCCL all-gather of the boundary block layers (device tensors) once, before marching cubes"}[args.exchange]
                                  if world > 1 else "none at N = 1"),
                   "blocks_per_rank_max": int(mx[3]), "blocks_per_rank_min": int(mn[3]), "blocks_total": int(sm[3]), "alloc_failures": int(sm[7])},
        "exchange": {"seconds": round(float(mx[1]), 4), "boundary_blocks_sent_total": int(sm[4]), "ghost_blocks_received_total": int(sm[5]),
                     "mode": args.exchange, "payload_bytes_received_per_rank_max": int(mx[8]), "payload_bytes_received_total": int(sm[8]),
                     "all_gather_would_receive_per_rank": int(sm[4]) * 4108},
        "marching_cubes": {"seconds": round(float(mx[2]), 4), "faces_total": int(sm[6])},
        "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else (0 if world > 1 else 1),
        "process_group": (dist.get_backend() if world > 1 else "none (one process)"),
        "per_rank_blocks": None if world == 1 else {"max": int(mx[3]), "min": int(mn[3])},
        "prefix_check": prefix,
        "roofline": None,
    }


def partition_prefix_check(args, frames, stride, poses, params, rank, local_rank, world, dist, n=96):
    """The N-rank path against ONE fuser on a bounded prefix: every rank fuses the first n frames with its stripes, the boundary layers travel, every
    rank meshes its own blocks, rank 0 merges the meshes by key -- and fuses the same n frames alone, without a partition.  The two canonical
    meshes (vertices in edge-key order, triangles in cube-key order) must be byte-identical."""
    import hashlib
    from scannet_amd import fusion, partition
    n = min(n, len(poses))
    with fusion.Fuser(params, device=local_rank, **TUNE) as f:
        if args.partition == "stripes":
            f.set_stripes(0, 0, args.stripe_blocks, world, rank)
        else:
            from scannet_amd import synth
            planes = partition.slab_planes(0, int(np.ceil(synth.CORRIDOR_ROOMS * synth.CORRIDOR_PITCH / (8 * params.voxel_size))), world)
            f.set_slab(0, planes[rank], planes[rank + 1])
        f.integrate_batch_device(frames.data_ptr(), stride, poses[:n])
        f.sync()
        sent, got = partition.exchange_boundary(f, mode=args.exchange)
        m = f.extract_mesh()
        xyz, rgba, tris, keys = m.arrays(keys=True)
        part = (xyz, rgba, tris, keys, m.face_keys())
    parts = [None] * world
    dist.all_gather_object(parts, part)
    counts = [None] * world
    dist.all_gather_object(counts, (int(sent), int(got)))
    if rank != 0:
        return None

    def sha(xyz, rgba, tris):
        return hashlib.sha256(np.ascontiguousarray(xyz).tobytes() + np.ascontiguousarray(rgba).tobytes() + np.ascontiguousarray(tris).tobytes()).hexdigest()

    mx_, mr_, mt_, _ = partition.merge_slab_meshes(parts)
    with fusion.Fuser(params, device=local_rank, **TUNE) as f:
        f.integrate_batch_device(frames.data_ptr(), stride, poses[:n])
        ox, orgba, ot, _ = f.extract_mesh().arrays(keys=True)
    h_n, h_1 = sha(mx_, mr_, mt_), sha(ox, orgba, ot)
    return {"frames": n, "faces": int(len(ot)), "merged_mesh_sha256": h_n[:16], "one_fuser_mesh_sha256": h_1[:16], "sha256_equal": h_n == h_1,
            "boundary_blocks_sent": sum(c[0] for c in counts), "ghost_blocks_received": sum(c[1] for c in counts),
            "what": "first %d frames: %d ranks with their stripes -> boundary exchange (%s) -> marching cubes per rank -> merge by key, against one fuser "
                    "without a partition on rank 0: canonical vertex / colour / triangle arrays hashed" % (n, world, args.exchange)}


COMPACT_LIMIT = 4000   # bytes: the driver keeps an 8 KB tail of stdout; round 5's 24 KB line could not be parsed from it


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, detail_path=None):
    """The one line the driver parses: the contract's keys, `roofline` and `cpu_baseline` reduced to numbers (no prose), the parity verdict and the
    end-to-end rates.  Optional groups are dropped from the end until the line fits COMPACT_LIMIT; the contract keys never are."""
    c = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c["vs_baseline"] = out.get("vs_baseline")
    cfg = out.get("config") or {}
    wl = str(cfg.get("workload", ""))
    c["config"] = dict({"workload": wl if len(wl) <= 260 else wl[:257] + "..."},
                       **_pick(cfg, "rgbd", "frames_per_pass", "integrate_launches", "blocks_live_end", "alloc_failures", "sharding", "scans", "partition", "exchange", "host_stage", "tune"))
    if isinstance(c["config"].get("sharding"), str) and len(c["config"]["sharding"]) > 80:
        c["config"]["sharding"] = c["config"]["sharding"][:77] + "..."
    r = out.get("roofline")
    if isinstance(r, dict):
        rr = _pick(r, "bound", "frac", "achieved", "peak", "kernel", "avg_kernel_us", "frames_per_launch", "traffic", "hbm_frac", "valu_frac_2cycle",
                   "frac_class_priced", "ta_busy", "tcp_tag_lookups_per_cu_clk", "traffic_over_must_move", "alg_bytes_per_launch")
        rr["unit"] = {"valu": "VALU issue fraction (SQ_INSTS_VALU x 2 clk / SIMD clk)", "hbm": "GB/s"}.get(r.get("bound"), str(r.get("unit"))[:60])
        for k in ("bound", "frac", "achieved", "peak", "traffic"):   # the contract's keys are always there (null: not measured in this run)
            rr.setdefault(k, r.get(k))
        if isinstance(rr.get("kernel"), str) and len(rr["kernel"]) > 64:
            rr["kernel"] = rr["kernel"][:61] + "..."
        hb = r.get("hbm_alg_batch")
        if isinstance(hb, dict):
            rr["hbm_must_move"] = _pick(hb, "bytes_per_launch", "GBs", "frac")
        ooc = r.get("hbm_out_of_cache")
        if isinstance(ooc, dict):
            rr["hbm_out_of_cache"] = _pick(ooc, "frac", "achieved_GBs", "kernel_alone_frac", "frac_of_rmw_ceiling")
        fc = r.get("front_chain")
        if isinstance(fc, dict):
            rr["front_chain_us_alone"] = {k: e.get("avg_us_alone") for k, e in fc.items() if isinstance(e, dict)}
        c["roofline"] = rr
    else:
        c["roofline"] = None
    b = out.get("cpu_baseline")
    if isinstance(b, dict):
        bb = _pick(b, "value", "unit", "cores", "kind", "frames")
        if isinstance(b.get("by_threads"), dict):   # threads -> frames/s (reference decode included where it was timed)
            bb["by_threads"] = {k: (e.get("frames_per_s_with_reference_decode", e.get("frames_per_s_integrate")) if isinstance(e, dict) else e) for k, e in b["by_threads"].items()}
        s_ = str(b.get("sample", ""))
        bb["sample"] = s_ if len(s_) <= 160 else s_[:157] + "..."
        c["cpu_baseline"] = bb
        if isinstance(b.get("value"), (int, float)) and b["value"] > 0 and isinstance(out.get("value"), (int, float)):
            c["gpu_over_cpu"] = round(out["value"] / b["value"], 1)
    optional = []
    for key in ("parity", "parity_depth_only"):
        pz = out.get(key)
        if isinstance(pz, dict):
            optional.append((key, _pick(pz, "sha256_equal", "frames", "blocks", "bit_exact")))
    for key in ("end_to_end", "end_to_end_rgbd"):
        e = out.get(key)
        if isinstance(e, dict):
            ee = _pick(e, "frames_per_s", "frames_per_s_best", "frames", "colour_fused", "decode_threads", "error")
            if isinstance(e.get("page_cache"), str):
                ee["page_cache"] = e["page_cache"].split(":")[0]
            if isinstance(e.get("depth_only"), dict):
                ee["depth_only"] = _pick(e["depth_only"], "frames_per_s", "frames_per_s_best", "frames", "error")
            optional.append((key, ee))
    for key in ("value_depth_only", "per_rank_frames_per_s", "rccl_ranks", "process_group"):
        if out.get(key) is not None:
            optional.append((key, out[key]))
    r1 = out.get("roofline_single_frame")
    if isinstance(r1, dict):
        optional.append(("roofline_single_frame", _pick(r1, "bound", "frac", "achieved", "avg_kernel_us", "frames_per_s", "traffic")))
    for k, v in optional:
        c[k] = v
    c["detail"] = detail_path
    line = json.dumps(c, separators=(",", ":"))
    while len(line) > COMPACT_LIMIT and optional:
        k, _ = optional.pop()
        c.pop(k, None)
        line = json.dumps(c, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:   # still too long: the free-text fields go, the numbers stay
        c["config"] = {"workload": c["config"]["workload"][:120]}
        line = json.dumps(c, separators=(",", ":"))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["4mm", "1mm", "scans", "partition"], default="4mm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the integrate kernel with HIP events")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes (VALU utilisation, HBM traffic)")
    ap.add_argument("--pmc-steps", type=int, default=None)
    ap.add_argument("--single-frame", action="store_true", help="one frame per launch (batch = 1) for the main measurement")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the secondary one-frame-per-launch roofline pass")
    ap.add_argument("--depth-only", action="store_true", help="time the geometry-only path (no colour frames): the headline of rounds 1-3")
    ap.add_argument("--no-depth-only", action="store_true", help="skip the secondary geometry-only pass (value_depth_only, roofline_depth_only)")
    ap.add_argument("--no-colour", action="store_true", help="accepted for old command lines; the colour path is the default measurement now (see --depth-only)")
    ap.add_argument("--no-e2e-rgbd", action="store_true", help="skip the end-to-end leg with 1296x968 JPEG colour")
    ap.add_argument("--no-prefix-check", action="store_true", help="--config partition: skip the merged-mesh check of a bounded prefix against one fuser")
    ap.add_argument("--teardown", action="store_true", help="leave through the interpreter's normal teardown (set for the runs under rocprofv3)")
    ap.add_argument("--child", action="store_true", help="a counter pass of another bench.py: the K timed steps once, nothing else")
    ap.add_argument("--repeats", type=int, default=0, help="how often the timed steps are repeated on an emptied volume (default: until ~1 s of timed GPU time)")
    ap.add_argument("--scene", type=int, choices=[0, 1], default=1, help="synthetic scene: 0 = the empty box room of rounds 1-2, 1 = the furnished room (default)")
    ap.add_argument("--noise", type=int, choices=[0, 1, 2], default=2, help="depth noise: 1 = the LCG ramp of rounds 1-2, 2 = three LSBs hashed per pixel and frame (default)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (.sens in /tmp -> sf_fuse_run)")
    ap.add_argument("--e2e-frames", type=int, default=5578, help="frames of the end-to-end leg (.sens in /tmp -> sf_fuse_run): the whole scene0000_00-scale scan by default")
    ap.add_argument("--e2e-rgbd-frames", type=int, default=5578, help="frames of the RGB-D end-to-end leg (1296x968 JPEG colour over zlib depth): the whole scan by default")
    ap.add_argument("--cpu-frames", type=int, default=200, help="frames of the CPU-baseline / parity leg")
    ap.add_argument("--no-out-of-cache", action="store_true", help="skip the bounded 1 mm (configs[2]) sub-measurement")
    ap.add_argument("--host-stage", choices=["full", "gpu-decimate", "gpu", "clean", "none"], default="gpu",
                    help="--config scans: what follows marching cubes -- gpu (default): clean.mlx, quadric collapse x 2 and cleanLoRes on the GPU + segment on host "
                         "threads; full: clean + SEQUENTIAL quadric collapse x 2 + segment on host threads (GPU idle 86 %%); gpu-decimate: the host chain with only "
                         "the collapse on the GPU (sf_mesh_simplify_gpu); clean: clean + segment; none: nothing")
    ap.add_argument("--first-scan", type=int, default=0)
    ap.add_argument("--max-scan-frames", type=int, default=0)
    ap.add_argument("--scan-frames", type=int, default=50000, help="--config partition: length of the long scan")
    ap.add_argument("--partition", choices=["stripes", "slabs"], default="stripes")
    ap.add_argument("--stripe-blocks", type=int, default=16)
    ap.add_argument("--exchange", choices=["neighbour", "all_gather"], default="neighbour", help="--config partition: how the boundary layers travel")
    ap.add_argument("--stripes-at-one", action="store_true", help="set the partition even at N = 1 (rank 0 of 1 owns everything)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="sf_fuser_tune switches for every fuser of the run (A/B measurements)")
    ap.add_argument("--share-gpu", action="store_true", help="testing aid: every rank on GPU 0 over the gloo backend (RCCL refuses two ranks on one device) -- "
                                                             "exercises the N > 1 control flow (barriers, max over ranks) on a one-GPU box; the rates mean nothing")
    args = ap.parse_args()
    if args.pmc_steps is None:
        args.pmc_steps = 400 if args.config == "4mm" else 64
    TUNE.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune})

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a bare `python bench.py --gpus N`: become the launcher -- one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1
        # (--gpus was parsed and never read; a bare invocation measured ONE GPU and said so only in n_gpus)
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not args.share_gpu:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); refusing to measure fewer ranks than asked for "
                             "(--share-gpu runs N ranks on GPU 0 over gloo to exercise the control flow)" % (args.gpus, ndev))
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a rate for a different number of GPUs"
                         % (args.gpus, world))
    if not args.share_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks on a node with %d GPU(s) (--share-gpu puts them all on GPU 0 over gloo)" % (world, torch.cuda.device_count()))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))

    if args.config in CONFIGS:
        out = run_stream(args, args.config, rank, local_rank, world, dist, torch)
    elif args.config == "scans":
        out = run_scans(args, rank, local_rank, world, dist, torch)
    else:
        out = run_partition(args, rank, local_rank, world, dist, torch)
    if rank == 0 and out is not None:
        # the record: ONE compact line (< 4 KB) as the last line of stdout; everything else of the measurement goes to bench_detail.json
        if args.child:   # a counter pass of another bench.py: the parent reads the whole measurement from this line
            print(json.dumps(out))
            sys.stdout.flush()
            out = None
    if rank == 0 and out is not None:
        detail = os.environ.get("SF_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(detail, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
        except OSError as ex:
            detail = "not written (%s)" % ex
        print(compact_line(out, detail))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # the line is out and every handle is closed: leave without the interpreter's teardown (HIP, RCCL and the OpenMP runtime of the CPU
    # port each register exit handlers, and their order is nobody's contract)
    sys.stdout.flush()
    sys.stderr.flush()
    under_profiler = args.teardown or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if not under_profiler:   # a profiler writes its results from exactly those exit handlers
        os._exit(0)


if __name__ == "__main__":
    main()
