#!/usr/bin/env python3
"""Run the PMC calibration stream (scannet_amd/csrc/calib.hip) under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE:
a known-byte-count 16 B/lane read-modify-write and read-only pass, once at 230 MB (fits the 256 MiB Infinity
Cache, like one frame's working set) and once at 2 GiB (does not)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannet_amd import _abi  # noqa: E402

L = _abi.lib()
L.sf_calib_stream.argtypes = [C.c_int, C.c_uint64, C.c_int]
for nbytes in (230 * 1000 * 1000 // 16 * 16, 2 << 30):
    _abi.check(L.sf_calib_stream(0, nbytes, 4))
    print("calib", nbytes)
