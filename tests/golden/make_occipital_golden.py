#!/usr/bin/env python3
"""Golden digests for tests/test_occipital.py, produced by the REFERENCE codec compiled from /root/reference
(oracle/_ref/libref_occ.so: ScannerApp/depth2pgm/uplinksimple_image-codecs.h + uplinksimple_shift2depth.h):
  table    sha256 of shift2depth(s) for s = 0..65535 as little-endian u16
  streams  sha256 over the reference ENCODER's output for the six synthetic shift frames of the test
Run in the container that holds /root/reference:  python tests/golden/make_occipital_golden.py > tests/golden/occipital_golden.json"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_occipital import _shift_frames  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_occ.so"))
L.ref_occ_shift2depth.argtypes = [C.c_uint16]
L.ref_occ_shift2depth.restype = C.c_uint16
L.ref_occ_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
L.ref_occ_encode.restype = C.c_uint32
table = np.array([L.ref_occ_shift2depth(s) for s in range(65536)], np.uint16)
h = hashlib.sha256()
sizes = []
for f in _shift_frames():
    a = np.ascontiguousarray(f.ravel())
    out = np.zeros(2 * a.size + 64, np.uint8)
    n = L.ref_occ_encode(a.ctypes.data, a.size, out.ctypes.data, out.size)
    h.update(out[:n].tobytes())
    sizes.append(int(n))
print(json.dumps({"generator": "tests/golden/make_occipital_golden.py (reference codec, oracle/_ref/libref_occ.so)",
                  "table": hashlib.sha256(table.tobytes()).hexdigest(), "streams": h.hexdigest(), "stream_bytes": sizes}, indent=1))
