import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scannet_amd import _abi, fusion, synth
W, H = 640, 480
N = 24
frames = torch.empty((N, H, W), dtype=torch.int16, device="cuda")
poses = synth.render_scan_device(frames.data_ptr(), W * H * 2, 0, N, 5578, W, H, noise=2, scene=1, seed=0)
L = _abi.lib()
L.sf_fuser_alloc_direct_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
L.sf_fuser_alloc_probe_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
import os
EXTRA = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("SF_PROBE_TUNE", "").split(",") if kv)}   # e.g. SF_PROBE_TUNE=overlap=0,pipe_wgs=2
BC = int(os.environ.get("SF_PROBE_BRICK_CACHE", "1"))   # the allocation kernels' presence cache on / off
for batch, gw in (((1, 1),) if os.environ.get("SF_PROBE_ONLY_BATCH1") else ((1, 1), (32, 1), (32, 4))):
    p = fusion.default_params(voxel_size=0.001, hash_num_buckets=1 << 22, num_sdf_blocks=1 << 24)
    with fusion.Fuser(p, batch=batch, alloc_group_win64=gw, brick_cache=BC, **EXTRA) as f:
        f.integrate_batch_device(frames[:8].data_ptr(), W * H * 2, poses[:8]); f.sync()
        n0 = C.c_uint64(0); _abi.check(L.sf_fuser_alloc_direct_count(f._h, C.byref(n0)))
        q0 = C.c_uint64(0); _abi.check(L.sf_fuser_alloc_probe_count(f._h, C.byref(q0)))
        t = time.perf_counter()
        f.integrate_batch_device(frames[8:].data_ptr(), W * H * 2, poses[8:]); f.sync()
        dt = time.perf_counter() - t
        n1 = C.c_uint64(0); _abi.check(L.sf_fuser_alloc_direct_count(f._h, C.byref(n1)))
        q1 = C.c_uint64(0); _abi.check(L.sf_fuser_alloc_probe_count(f._h, C.byref(q1)))
        st = f.stats()
        if hasattr(L, "sf_alloc_timing_read"):   # -DSF_ALLOC_TIMING builds: where the workgroups of k_alloc spent their time (100 MHz clock, per workgroup and launch)
            t16 = (C.c_uint64 * 16)(); L.sf_alloc_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]; _abi.check(L.sf_alloc_timing_read(f._h, t16))
            wg = max(1, t16[10]); us = lambda v: round(v / 100.0 / wg, 1)
            print("k_alloc timing per workgroup, us: setup", us(t16[0]), "anchor", us(t16[1]), "walk", us(t16[2]), "scan", us(t16[3]), "drain", us(t16[4]), "whole", us(t16[5]),
                  "| longest workgroup", round(t16[8] / 100.0, 1), "| rounds per workgroup", round(t16[9] / wg, 2), "| workgroups", t16[10])
            lg = (C.c_uint32 * (1 + 256 * 12))(); L.sf_alloc_timing_log.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]; _abi.check(L.sf_alloc_timing_log(f._h, lg))
            rows = sorted((list(lg[1 + 12 * i: 13 + 12 * i]) for i in range(min(256, lg[0]))), key=lambda r: -r[7])
            print("workgroups over 250 us:", lg[0], "(the 24 longest of the first 256: tile x y | rounds | setup anchor walk scan drain whole, us | direct probed queued)")
            for r in rows[:24]:
                print("   tile", r[0] & 0xffff, r[0] >> 16, "| rounds", r[1], "|", " ".join(str(round(v / 100.0, 1)) for v in r[2:8]), "|", r[8], r[9], r[10])
        print("brick_cache", BC, "table probes in 16 frames", q1.value - q0.value, end=" ")
        print("batch", batch, "alloc_group_win64", gw, "fps", round((N - 8) / dt, 1), "direct-path blocks in 16 frames", n1.value - n0.value, "blocks allocated", st["blocks_allocated"], "frame blocks", st["last_frame_blocks"])
