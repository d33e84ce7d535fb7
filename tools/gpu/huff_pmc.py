"""One 1296x968 picture through k_jpeg_huff a few times (sf_jpeg_decode_gpu_huffman): the workload for `rocprofv3 --pmc ... -- python tools/gpu/huff_pmc.py`."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scannet_amd import _abi, calibrate, synth   # noqa: E402

W, H = 1296, 968
pics = synth.textured_pictures(W, H, count=2)
L = _abi.lib()
L.sf_jpeg_decode_gpu_huffman.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
out = np.zeros((H, W, 3), np.uint8)
for rep in range(3):
    for p in pics:
        blob = p if isinstance(p, (bytes, bytearray)) else calibrate.jpeg_encode(np.asarray(p), 90, True)
        _abi.check(L.sf_jpeg_decode_gpu_huffman(blob, len(blob), W, H, 0, out.ctypes.data_as(C.c_void_p)))
print("decoded", len(blob), "bytes", int(out.sum()))
