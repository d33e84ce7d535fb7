"""Annotation projection -- host-side mirror of AnnotationTools/ProjectAnnotations (Visualizer::render and the vertex labelling of
Visualizer::computeObjectIdsAndColorsPerVertex / propagateAnnotations) over the C ABI; the rasteriser and the image filters run on
the GPU (scannet_amd/csrc/project.hip)."""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check


class ProjectParams(C.Structure):
    _fields_ = [("color_width", C.c_uint32), ("color_height", C.c_uint32), ("depth_width", C.c_uint32), ("depth_height", C.c_uint32),
                ("fx", C.c_float), ("fy", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float), ("depth_dist_thresh", C.c_float),
                ("filter_using_original_depth", C.c_int32)]


def default_params(color_wh, depth_wh, fx, fy, **kw):
    """zParametersScan.txt:6,12-14 defaults."""
    p = ProjectParams(color_wh[0], color_wh[1], depth_wh[0], depth_wh[1], fx, fy, 0.1, 15.0, 0.2, 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _lib():
    L = _abi.lib()
    vp, u64 = C.c_void_p, C.c_uint64
    L.sf_projector_create.argtypes = [C.POINTER(ProjectParams), C.c_int, C.POINTER(vp)]
    L.sf_projector_destroy.argtypes = [vp]
    L.sf_projector_destroy.restype = None
    L.sf_projector_max_batch.argtypes = []
    L.sf_projector_set_mesh.argtypes = [vp, vp, u64, vp, u64, vp, vp]
    L.sf_projector_run.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.sf_host_alloc.argtypes = [u64, C.POINTER(vp)]
    L.sf_host_free.argtypes = [vp]
    L.sf_host_free.restype = None
    L.sf_annotation_vertex_ids.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, u64, vp, vp, C.POINTER(C.c_uint32)]
    L.sf_annotation_propagate.argtypes = [vp, u64, vp, u64, vp, vp, vp, u64, vp, u64, C.c_float, vp, vp]
    return L


def vertex_ids(segs_json, aggregation_json, label_map_tsv, num_vertices):
    """Visualizer.cpp:259-295 -> (instance u8[V], label u16[V], number of labels)."""
    inst = np.zeros(num_vertices, np.uint8)
    label = np.zeros(num_vertices, np.uint16)
    n = C.c_uint32()
    check(_lib().sf_annotation_vertex_ids(str(segs_json).encode(), str(aggregation_json).encode(), str(label_map_tsv).encode(), num_vertices,
                                          inst.ctypes.data, label.ctypes.data, C.byref(n)))
    return inst, label, n.value


def propagate(src_xyz, src_tris, src_inst, src_label, dst_xyz, dst_tris, normal_thresh=0.5):
    """Visualizer.cpp:297-377 -> (instance u8[Vd], label u16[Vd])."""
    sx, st = np.ascontiguousarray(src_xyz, np.float32).reshape(-1, 3), np.ascontiguousarray(src_tris, np.uint32).reshape(-1, 3)
    dx, dt = np.ascontiguousarray(dst_xyz, np.float32).reshape(-1, 3), np.ascontiguousarray(dst_tris, np.uint32).reshape(-1, 3)
    si, sl = np.ascontiguousarray(src_inst, np.uint8), np.ascontiguousarray(src_label, np.uint16)
    assert si.size == len(sx) and sl.size == len(sx)
    di, dl = np.zeros(len(dx), np.uint8), np.zeros(len(dx), np.uint16)
    check(_lib().sf_annotation_propagate(sx.ctypes.data, len(sx), st.ctypes.data, len(st), si.ctypes.data, sl.ctypes.data, dx.ctypes.data, len(dx),
                                         dt.ctypes.data, len(dt), float(normal_thresh), di.ctypes.data, dl.ctypes.data))
    return di, dl


class Projector:
    def __init__(self, params, device=0):
        self._h = C.c_void_p()
        self.params = params
        check(_lib().sf_projector_create(C.byref(params), int(device), C.byref(self._h)))
        self.max_batch = _lib().sf_projector_max_batch()
        self._pin = {}

    def close(self):
        if self._h:
            _lib().sf_projector_destroy(self._h)
            self._h = None
            for ptr, _ in self._pin.values():
                _lib().sf_host_free(ptr)
            self._pin = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _pinned(self, name, shape, dtype):
        """a numpy view of page-locked memory, kept per (name, shape) for the life of the projector"""
        key = (name, tuple(shape), np.dtype(dtype).str)
        if key not in self._pin:
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            ptr = C.c_void_p()
            check(_lib().sf_host_alloc(nbytes, C.byref(ptr)))
            self._pin[key] = (ptr, np.frombuffer((C.c_uint8 * nbytes).from_address(ptr.value), dtype=dtype).reshape(shape))
        return self._pin[key][1]

    def set_mesh(self, xyz, tris, vertex_instance, vertex_label):
        x, t = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3), np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        i, l = np.ascontiguousarray(vertex_instance, np.uint8), np.ascontiguousarray(vertex_label, np.uint16)
        assert i.size == len(x) and l.size == len(x)
        check(_lib().sf_projector_set_mesh(self._h, x.ctypes.data, len(x), t.ctypes.data, len(t), i.ctypes.data, l.ctypes.data))

    def run(self, cam2world, orig_depth=None, want_depth=False, pinned=False):
        """cam2world (n, 4, 4); orig_depth (n, dh, dw) u16 or None -> (instance (n, ch, cw) u8, label u16[, rendered depth f32], kernel microseconds).
        pinned: the outputs are views of page-locked buffers owned by the projector, overwritten by the next call."""
        p = self.params
        c = np.ascontiguousarray(cam2world, np.float32).reshape(-1, 16)
        n = len(c)
        d = None
        if orig_depth is not None:
            d = np.ascontiguousarray(orig_depth, np.uint16).reshape(n, p.depth_height, p.depth_width)
        if pinned:
            inst = self._pinned("inst", (n, p.color_height, p.color_width), np.uint8)
            label = self._pinned("label", (n, p.color_height, p.color_width), np.uint16)
            if d is not None:
                dp = self._pinned("depth", d.shape, np.uint16)
                dp[...] = d
                d = dp
        else:
            inst = np.empty((n, p.color_height, p.color_width), np.uint8)
            label = np.empty((n, p.color_height, p.color_width), np.uint16)
        z = np.empty((n, p.color_height, p.color_width), np.float32) if want_depth else None
        us = C.c_float()
        check(_lib().sf_projector_run(self._h, n, c.ctypes.data, d.ctypes.data if d is not None else None, inst.ctypes.data, label.ctypes.data,
                                      z.ctypes.data if z is not None else None, C.byref(us)))
        return (inst, label, z, us.value) if want_depth else (inst, label, us.value)
