"""2-D annotation filter -- host-side mirror of AnnotationTools/Filter2dAnnotations (FilterData + the frame body of process())
over the C ABI; the kernels run on the GPU (scannet_amd/csrc/filter2d.hip)."""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check

MAX_NUM_LABELS_PER_SCENE = 80  # GlobalDefines.h:12


def _lib():
    L = _abi.lib()
    vp = C.c_void_p
    L.sf_filter2d_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.sf_filter2d_destroy.argtypes = [vp]
    L.sf_filter2d_destroy.restype = None
    L.sf_filter2d_set_tables.argtypes = [vp, vp, vp, vp]
    L.sf_filter2d_frame.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    return L


def make_tables(object_ids_to_label):
    """Filter2dAnnotations.cpp:293-309: {object id (0-based): label id} -> (instance_to_idx[256], idx_to_instance[80], instance_to_label[256])."""
    to_idx = np.full(256, 255, np.uint8)
    to_inst = np.full(MAX_NUM_LABELS_PER_SCENE, 255, np.uint8)
    to_label = np.full(256, 65535, np.uint16)
    to_idx[0] = 0
    to_inst[0] = 0
    to_label[0] = 0
    idx = 1
    for obj, label in sorted(object_ids_to_label.items()):
        to_label[obj + 1] = label
        to_idx[obj + 1] = idx
        to_inst[idx] = obj + 1
        idx += 1
    return to_idx, to_inst, to_label


class Filter2d:
    def __init__(self, depth_wh, color_wh, device=0):
        self._h = C.c_void_p()
        self.depth_wh, self.color_wh = depth_wh, color_wh
        check(_lib().sf_filter2d_create(depth_wh[0], depth_wh[1], color_wh[0], color_wh[1], int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            _lib().sf_filter2d_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_tables(self, to_idx, to_inst, to_label):
        a, b, c = (np.ascontiguousarray(to_idx, np.uint8), np.ascontiguousarray(to_inst, np.uint8), np.ascontiguousarray(to_label, np.uint16))
        assert a.size == 256 and b.size == 80 and c.size == 256
        self._keep = (a, b, c)
        check(_lib().sf_filter2d_set_tables(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data))

    def frame(self, depth, rgb, instance):
        """-> (instance_out uint8 [Hc, Wc], label_out uint16 [Hc, Wc], kernel microseconds)"""
        d = np.ascontiguousarray(depth, np.uint16)
        c = np.ascontiguousarray(rgb, np.uint8)
        i = np.ascontiguousarray(instance, np.uint8)
        cw, ch = self.color_wh
        io = np.empty((ch, cw), np.uint8)
        lo = np.empty((ch, cw), np.uint16)
        us = C.c_float(0)
        check(_lib().sf_filter2d_frame(self._h, d.ctypes.data, c.ctypes.data, i.ctypes.data, io.ctypes.data, lo.ctypes.data, C.byref(us)))
        return io, lo, us.value


# ---- PNG images of the annotation tools (scannet_amd/csrc/png.cpp) -------------------------------------------------------
def png_read(path):
    """-> uint8 / uint16 array [H, W] (grey) or [H, W, C]."""
    import os
    L = _abi.lib()
    L.sf_png_read.argtypes = [C.c_char_p] + [C.POINTER(C.c_uint32)] * 2 + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_void_p)]
    L.sf_free.argtypes = [C.c_void_p]
    L.sf_free.restype = None
    w, h, c, b, d = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int(), C.c_void_p()
    check(L.sf_png_read(os.fsencode(path), C.byref(w), C.byref(h), C.byref(c), C.byref(b), C.byref(d)))
    ct = C.c_uint16 if b.value == 16 else C.c_uint8
    a = np.ctypeslib.as_array(C.cast(d, C.POINTER(ct)), (h.value, w.value, c.value)).copy()
    L.sf_free(d)
    return a[..., 0] if c.value == 1 else a


def png_write_gray(path, image):
    import os
    a = np.ascontiguousarray(image)
    assert a.ndim == 2 and a.dtype in (np.uint8, np.uint16)
    L = _abi.lib()
    L.sf_png_write_gray.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    check(L.sf_png_write_gray(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], 8 * a.dtype.itemsize))
