#!/usr/bin/env python3
"""Calibrate-stage throughput on one MI355X (numbers for DESIGN.md; not bench.py's metric): 16 frames per call, inputs resident
in HBM, four launches per call timed with HIP events on the calibrator's stream; the CPU checker (oracle/calib_oracle.c, one
core -- the reference runs this body under OpenMP with the warp serialised in an `omp critical`) on a bounded sample beside it.

  python tools/calib_bench.py [--iters 50] [--out gpurun_out/calib.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannet_amd import _abi, calibrate  # noqa: E402
from tests import test_calibrate as T  # noqa: E402  (scene generator shared with the parity tests)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    W, H, CW, CH = T.W, T.H, T.CW, T.CH
    B = 16
    p = T._full_params()
    grid, maxd = T._lut()
    frames = [T._scene(frame=10 * i, seed=i) for i in range(B)]
    depth = np.stack([f[0] for f in frames])
    rgb = np.stack([f[1] for f in frames])
    L = calibrate._lib()
    Labi = _abi.lib()
    res = {"frames_per_call": B, "depth": [W, H], "color": [CW, CH]}
    with calibrate.Calibrator(p, grid, maxd) as cal:
        def dev(buf):
            ptr = C.c_void_p()
            _abi.check(Labi.sf_device_malloc(0, buf.nbytes, C.byref(ptr)))
            _abi.check(Labi.sf_device_upload(ptr, buf.ctypes.data, buf.nbytes))
            return ptr
        d_depth_in, d_rgb_in = dev(depth), dev(rgb)
        d_depth_out, d_rgb_out = dev(np.zeros_like(depth)), dev(np.zeros_like(rgb))
        vp = C.c_void_p
        nb, cb = W * H * 2, CW * CH * 3
        arr = lambda base, stride: (vp * B)(*[base.value + i * stride for i in range(B)])
        di, do, ri, ro = arr(d_depth_in, nb), arr(d_depth_out, nb), arr(d_rgb_in, cb), arr(d_rgb_out, cb)
        for with_rgb in (True, False):
            us = C.c_float(0)
            tot = 0.0
            for it in range(a.iters + 3):
                _abi.check(L.sf_calibrator_run_device(cal._h, B, ri if with_rgb else None, ro if with_rgb else None, di, do, C.byref(us)))
                if it >= 3:
                    tot += us.value
            per_frame_us = tot / a.iters / B
            # bytes every frame must move at least once: colour in + out, depth in, metres out + in (gather), depth buffer clear +
            # read, depth out (atomic traffic on the depth buffer is extra)
            alg = (2 * cb if with_rgb else 0) + W * H * (2 + 4 + 4 + 4 + 4 + 2)
            res["with_colour" if with_rgb else "depth_only"] = {
                "kernel_us_per_frame": round(per_frame_us, 3), "frames_per_s_kernels": round(1e6 / per_frame_us, 1),
                "alg_bytes_per_frame": alg, "achieved_GBs": round(alg / per_frame_us / 1e3, 1), "hbm_peak_GBs": 8000.0,
                "frac": round(alg / per_frame_us / 1e3 / 8000.0, 4)}
        for ptr in (d_depth_in, d_rgb_in, d_depth_out, d_rgb_out):
            Labi.sf_device_free(ptr)
        # end to end through host buffers (H2D + kernels + D2H, synchronous)
        t0 = time.perf_counter()
        n_calls = 10
        for _ in range(n_calls):
            cal.run(depth, rgb)
        dt = time.perf_counter() - t0
        res["host_buffers_frames_per_s"] = round(n_calls * B / dt, 1)
    from oracle import oracle as orc
    cbo = orc.calib_from(p)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 10.0 and n < 64:
        orc.calib_frame(cbo, depth[n % B], rgb[n % B], grid, maxd)
        n += 1
    dt = time.perf_counter() - t0
    res["cpu_baseline"] = {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                           "sample": "%d frames, oracle/calib_oracle.c -O2, %.1f s" % (n, dt)}
    s = json.dumps(res)
    print(s)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
