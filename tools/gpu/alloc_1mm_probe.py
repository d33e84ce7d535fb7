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
import os
for batch, gw in (((1, 1),) if os.environ.get("SF_PROBE_ONLY_BATCH1") else ((1, 1), (32, 1), (32, 4))):
    p = fusion.default_params(voxel_size=0.001, hash_num_buckets=1 << 22, num_sdf_blocks=1 << 24)
    with fusion.Fuser(p, batch=batch, alloc_group_win64=gw) as f:
        f.integrate_batch_device(frames[:8].data_ptr(), W * H * 2, poses[:8]); f.sync()
        n0 = C.c_uint64(0); _abi.check(L.sf_fuser_alloc_direct_count(f._h, C.byref(n0)))
        t = time.perf_counter()
        f.integrate_batch_device(frames[8:].data_ptr(), W * H * 2, poses[8:]); f.sync()
        dt = time.perf_counter() - t
        n1 = C.c_uint64(0); _abi.check(L.sf_fuser_alloc_direct_count(f._h, C.byref(n1)))
        st = f.stats()
        print("batch", batch, "alloc_group_win64", gw, "fps", round((N - 8) / dt, 1), "direct-path blocks in 16 frames", n1.value - n0.value, "blocks allocated", st["blocks_allocated"], "frame blocks", st["last_frame_blocks"])
