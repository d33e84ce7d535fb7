#!/usr/bin/env python3
"""Builds and runs the two study tools on three depth frames of the bench stream (the furnished room with hashed noise at two places of the walk, and
the round-1/2 input: the empty room with the LCG ramp), deflated by this library's writer (one fixed-Huffman block, as the reference's stb writer).
    python tools/inflate_parallelism/run.py > profiles/r04_inflate_parallelism.txt"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from scannet_amd import _abi, synth  # noqa: E402


def deflate(raw):
    L = _abi.lib()
    L.sf_zlib_deflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sf_zlib_deflate_bound.restype = C.c_uint64
    cap = L.sf_zlib_deflate_bound(C.c_uint64(len(raw)))
    out = np.zeros(cap, np.uint8)
    n = C.c_uint64(0)
    assert L.sf_zlib_deflate(raw.ctypes.data, len(raw), out.ctypes.data, cap, C.byref(n)) == 0
    return out[:n.value].tobytes()


def main():
    d = tempfile.mkdtemp(prefix="sf_infl_")
    for tool in ("emulate", "dependency_depth"):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-o", os.path.join(d, tool), os.path.join(HERE, tool + ".cpp")])
    for frame, noise, furnished, label in ((100, 2, True, "furnished room, hashed noise, frame 100"), (1400, 2, True, "furnished room, hashed noise, frame 1400"),
                                           (100, 1, False, "empty room, LCG ramp (the round-1/2 input), frame 100")):
        pose = synth.trajectory_pose(frame, 5578)
        depth = synth.render_room_depth(pose, 640, 480, noise_frame=frame, noise=noise, boxes=synth.clutter_boxes() if furnished else None)
        raw = depth.view(np.uint8).reshape(-1)
        blob = deflate(raw)
        path = os.path.join(d, "f.z")
        open(path, "wb").write(blob)
        print("== %s: %d bytes deflated (%d raw)" % (label, len(blob), len(raw)))
        sys.stdout.flush()
        subprocess.check_call([os.path.join(d, "dependency_depth"), path])
        subprocess.check_call([os.path.join(d, "emulate"), path, str(len(raw)), os.path.join(d, "f.out")])
        assert open(os.path.join(d, "f.out"), "rb").read() == zlib.decompress(blob) == raw.tobytes()
        print("   (lane emulation output identical to zlib's)")


if __name__ == "__main__":
    main()
