#!/usr/bin/env python3
"""Timing of the annotation projection on one MI355X at ScanNet's sizes: 1296x968 render target over 640x480 depth,
a room mesh of ~1.2 M triangles (a `_vh_clean.ply` is 1-4 M), frames in batches of eight.  Not bench.py's metric.

  python tools/project_bench.py [--frames 64] [--grid 245] [--out gpurun_out/project.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (the checker, timed beside the GPU path on one frame)
from scannet_amd import project  # noqa: E402
from tests.test_project import _pose, room_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--grid", type=int, default=245)
    ap.add_argument("--out", default="")
    ap.add_argument("--blind", action="store_true", help="move the cameras 100 m away looking outwards: every triangle is culled (per-triangle overhead only)")
    a = ap.parse_args()
    P = project.default_params((1296, 968), (640, 480), 1170.0, 1170.0)
    xyz, tris, inst, label = room_scene(a.grid, seed=2)
    t = np.linspace(0, 2 * np.pi, a.frames, endpoint=False)
    poses = np.stack([_pose([3.0 + 1.5 * np.cos(x), 2.0 + 0.8 * np.sin(x), 1.4], x + 1.6, -0.3) for x in t])
    if a.blind:
        poses[:, 0, 3] += 100.0
    depth = np.full((a.frames, 480, 640), 2000, np.uint16)
    res = {"vertices": len(xyz), "triangles": len(tris), "frames": a.frames, "color": [1296, 968], "depth": [640, 480]}
    with project.Projector(P) as pr:
        pr.set_mesh(xyz, tris, inst, label)
        B = pr.max_batch
        pr.run(poses[:B], depth[:B])
        t0 = time.perf_counter()
        kus = 0.0
        cover = 0.0
        for b in range(0, a.frames, B):
            gi, gl, us = pr.run(poses[b:b + B], depth[b:b + B])
            kus += us
            cover += float((gl != 0).mean()) * len(gl)
        wall = time.perf_counter() - t0
        res["gpu_kernel_us_per_frame"] = round(kus / a.frames, 1)
        res["wall_ms_per_frame_incl_copies_pageable"] = round(1e3 * wall / a.frames, 3)
        pr.run(poses[:B], depth[:B], pinned=True)
        t0 = time.perf_counter()
        for b in range(0, a.frames, B):
            pr.run(poses[b:b + B], depth[b:b + B], pinned=True)
        res["wall_ms_per_frame_incl_copies_pinned"] = round(1e3 * (time.perf_counter() - t0) / a.frames, 3)
        res["labelled_fraction"] = round(cover / a.frames, 3)
        g1 = pr.run(poses[:1], depth[:1], want_depth=True)
    t0 = time.perf_counter()
    oi, ol, oz = orc.project_frame(P, xyz, tris, inst, label, poses[0], depth[0], want_depth=True)
    res["cpu_checker_ms_per_frame_1_thread"] = round(1e3 * (time.perf_counter() - t0), 1)
    res["bit_exact_vs_checker"] = bool(np.array_equal(g1[0][0], oi) and np.array_equal(g1[1][0], ol) and np.array_equal(g1[2][0], oz))
    print(json.dumps(res))
    if a.out:
        open(a.out, "w").write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
