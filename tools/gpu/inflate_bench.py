#!/usr/bin/env python3
"""The device's inflate alone: 32 depth frames of the bench stream (this library's writer: one fixed-Huffman block, as the reference's), resident,
the two kernels timed apart.    python tools/gpu/inflate_bench.py [--frames 32] [--repeats 20]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scannet_amd import _abi, sens, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--repeats", type=int, default=20)
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--step", type=int, default=150)
    a = ap.parse_args()
    boxes = synth.clutter_boxes()
    blobs = []
    for k in range(a.frames):
        i = a.first + k * a.step
        d = synth.render_room_depth(synth.trajectory_pose(i, 5578), 640, 480, noise_frame=i, noise=2, boxes=boxes)
        blobs.append(np.frombuffer(sens.zlib_deflate(d.tobytes()), np.uint8))
    L = _abi.lib()
    L.sf_zlib_inflate_gpu_bench.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ptrs = (C.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    sizes = (C.c_uint64 * len(blobs))(*[b.size for b in blobs])
    t, c = C.c_double(0), C.c_double(0)
    parts = {}
    for skip, name in ((1, "us_tokens_without_the_writing_pass"), (2, "us_tokens_without_scans_and_writing_pass")):
        # the stage switches are compiled only with SCANFUSE_BUILD_FLAGS=-DSF_MEASURE_ABLATE (python -c 'from scannet_amd import build; build.build(force=True)')
        if L.sf_zlib_inflate_gpu_bench(ptrs, sizes, len(blobs), 614400, 0, a.repeats, skip, C.byref(t), C.byref(c)) == 0:
            parts[name] = round(t.value, 1)
    _abi.check(L.sf_zlib_inflate_gpu_bench(ptrs, sizes, len(blobs), 614400, 0, a.repeats, 0, C.byref(t), C.byref(c)))
    comp = sum(b.size for b in blobs)
    print(json.dumps({"frames": len(blobs), "compressed_bytes_per_frame": comp // len(blobs), "us_tokens": round(t.value, 1), **parts, "us_copy": round(c.value, 1),
                      "frames_per_s_if_serial": round(len(blobs) / ((t.value + c.value) * 1e-6)), "frames_per_s_if_overlapped": round(len(blobs) / (max(t.value, c.value) * 1e-6)),
                      "compressed_GBs_tokens": round(comp / (t.value * 1e-6) / 1e9, 2), "output_GBs_copy": round(len(blobs) * 614400 / (c.value * 1e-6) / 1e9, 2)}))


if __name__ == "__main__":
    main()
