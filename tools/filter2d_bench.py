#!/usr/bin/env python3
"""2-D annotation filter at the reference's image sizes (1296x968 colour, 640x480 depth) on one MI355X: duration of a frame's
kernels (HIP events on the filter's stream), the CPU checker beside it on ONE frame (OpenMP over rows, all host cores), and a
bit-exactness check of that frame.  Numbers for DESIGN.md; not bench.py's metric.

  python tools/filter2d_bench.py [--iters 5] [--out gpurun_out/filter2d.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannet_amd import filter2d  # noqa: E402


def scene(dw, dh, cw, ch, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:dh, 0:dw]
    depth = (1500 + 2 * xx + np.where(xx > dw // 2, 700, 0) + np.where(yy > dh * 2 // 3, 300, 0) + rng.integers(0, 4, (dh, dw))).astype(np.uint16)
    depth[rng.random((dh, dw)) < 0.01] = 0
    cy, cx = np.mgrid[0:ch, 0:cw]
    right = cx > cw // 2
    rgb = np.where(right[..., None], np.array([200, 60, 40]), np.array([40, 90, 200])).astype(np.int32) + rng.integers(-12, 13, (ch, cw, 3))
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    inst = np.where(cx + (20 * np.sin(cy / 25.0)).astype(int) > cw // 2, 2, 1).astype(np.uint8)
    inst[cy > ch * 2 // 3] = 3
    inst[cy < 40] = 0
    return depth, rgb, inst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    DW, DH, CW, CH = 640, 480, 1296, 968
    depth, rgb, inst = scene(DW, DH, CW, CH)
    tables = filter2d.make_tables({0: 4, 1: 7, 2: 39})
    res = {"depth": [DW, DH], "color": [CW, CH]}
    with filter2d.Filter2d((DW, DH), (CW, CH)) as f:
        f.set_tables(*tables)
        f.frame(depth, rgb, inst)
        us = []
        t0 = time.perf_counter()
        for _ in range(a.iters):
            io, lo, k = f.frame(depth, rgb, inst)
            us.append(k)
        wall = (time.perf_counter() - t0) / a.iters
        taps = CW * CH * (25 * 25 + 21 * 21) + DW * DH * 9 * 9 + 320 * 240 * 25 * 25
        res["gpu"] = {"kernel_ms_per_frame": round(float(np.mean(us)) / 1e3, 3), "wall_ms_per_frame_host_buffers": round(wall * 1e3, 3),
                      "window_taps_per_frame": taps, "taps_per_s": round(taps / (float(np.mean(us)) * 1e-6), 0)}
    if not a.no_cpu:
        from scannet_amd import _abi
        cpus = _abi.usable_cpus()   # the cgroup quota, not the logical CPUs the container shows
        os.environ.setdefault("OMP_NUM_THREADS", str(cpus))
        from oracle import oracle as orc
        t0 = time.perf_counter()
        oi, ol = orc.f2d_frame(depth, rgb, inst, *tables)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cpus, "kind": "port",
                               "sample": "1 frame, oracle/filter2d_oracle.c -O2 -fopenmp (%d threads, %d logical CPUs visible), %.1f s" % (cpus, os.cpu_count() or cpus, dt)}
        res["bit_exact_vs_checker"] = bool(np.array_equal(io, oi) and np.array_equal(lo, ol))
    s = json.dumps(res)
    print(s)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
