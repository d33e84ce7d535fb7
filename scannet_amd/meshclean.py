"""Mesh cleaning -- host-side mirror of the meshlabserver clean step (Server/scan_processor.py:134,143 with
Server/tools/meshclean/clean.mlx / cleanLoRes.mlx) over the C ABI."""
import ctypes as C
import os

from . import _abi
from ._abi import check
from .segmentator import Mesh

CLEAN_MLX_MERGE_DISTANCE = 0.0010689   # clean.mlx:4 (absolute)
CLEAN_MLX_MIN_COMPONENT = 7500         # clean.mlx:8
CLEAN_LORES_MIN_COMPONENT = 1000       # cleanLoRes.mlx:8


class SfCleanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("vertices_in", "faces_in", "vertices_merged", "faces_degenerate", "faces_duplicate",
                                          "components_in", "components_removed", "faces_small_component", "vertices_unreferenced",
                                          "vertices_out", "faces_out")]


class SfSimplifyParams(C.Structure):
    """simplify.mlx:3-16 ("Quadric Edge Collapse Decimation")."""
    _fields_ = [("target_faces", C.c_uint64), ("target_perc", C.c_float), ("quality_thr", C.c_float), ("preserve_boundary", C.c_int32),
                ("boundary_weight", C.c_float), ("preserve_normal", C.c_int32), ("preserve_topology", C.c_int32),
                ("optimal_placement", C.c_int32), ("planar_quadric", C.c_int32), ("quality_weight", C.c_int32), ("auto_clean", C.c_int32)]


class SfSimplifyStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("vertices_in", "faces_in", "target_faces", "collapses", "stale_popped", "faces_zero_area",
                                          "vertices_duplicate", "vertices_out", "faces_out")] + [("max_priority", C.c_float), ("rounds", C.c_uint32)]


class SfCleanScript(C.Structure):
    _fields_ = [("merge_close_vertices", C.c_int32), ("remove_duplicate_faces", C.c_int32), ("remove_small_components", C.c_int32),
                ("remove_unreferenced", C.c_int32), ("merge_distance", C.c_float), ("min_component_faces", C.c_uint32),
                ("simplify", C.c_int32), ("simplify_params", SfSimplifyParams), ("simplify_stats", SfSimplifyStats), ("simplify_device", C.c_int32), ("clean_device", C.c_int32)]


def _lib():
    L = _abi.lib()
    vp = C.c_void_p
    L.sf_mesh_clean.argtypes = [vp, C.c_float, C.c_uint32, C.POINTER(vp), C.POINTER(SfCleanStats)]
    L.sf_mlx_load.argtypes = [C.c_char_p, C.POINTER(SfCleanScript)]
    L.sf_mesh_clean_script.argtypes = [vp, C.POINTER(SfCleanScript), C.POINTER(vp), C.POINTER(SfCleanStats)]
    L.sf_simplify_default_params.argtypes = [C.POINTER(SfSimplifyParams)]
    L.sf_simplify_default_params.restype = None
    L.sf_mesh_simplify.argtypes = [vp, C.POINTER(SfSimplifyParams), C.POINTER(vp), C.POINTER(SfSimplifyStats)]
    return L


def simplify(mesh, gpu=None, **overrides):
    """"Quadric Edge Collapse Decimation" with simplify.mlx's parameters (keep 20 % of the faces) unless overridden;
    returns (Mesh, stats dict).  gpu=<device>: rounds of independent collapses on that GPU (sf_mesh_simplify_gpu) instead of the
    sequential host filter -- different triangles, same guarantees, a fraction of the time."""
    p = SfSimplifyParams()
    _lib().sf_simplify_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown simplify parameter %r" % k)
        setattr(p, k, v)
    h, st = C.c_void_p(), SfSimplifyStats()
    if gpu is None:
        check(_lib().sf_mesh_simplify(mesh._h, C.byref(p), C.byref(h), C.byref(st)))
    else:
        L = _lib()
        L.sf_mesh_simplify_gpu.argtypes = [C.c_void_p, C.POINTER(SfSimplifyParams), C.c_int, C.POINTER(C.c_void_p), C.POINTER(SfSimplifyStats)]
        check(L.sf_mesh_simplify_gpu(mesh._h, C.byref(p), int(gpu), C.byref(h), C.byref(st)))
    return Mesh(h), {n: getattr(st, n) for n, _ in SfSimplifyStats._fields_}


def clean(mesh, merge_distance=CLEAN_MLX_MERGE_DISTANCE, min_component_faces=CLEAN_MLX_MIN_COMPONENT, gpu=None):
    """The four clean.mlx filters on a Mesh; returns (Mesh, stats dict).  gpu=<device>: the same filters on that GPU (sf_mesh_clean_gpu),
    identical arrays and statistics."""
    h, st = C.c_void_p(), SfCleanStats()
    if gpu is None:
        check(_lib().sf_mesh_clean(mesh._h, float(merge_distance), int(min_component_faces), C.byref(h), C.byref(st)))
    else:
        L = _lib()
        L.sf_mesh_clean_gpu.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(SfCleanStats)]
        check(L.sf_mesh_clean_gpu(mesh._h, float(merge_distance), int(min_component_faces), int(gpu), C.byref(h), C.byref(st)))
    return Mesh(h), {n: getattr(st, n) for n, _ in SfCleanStats._fields_}


def load_script(path):
    s = SfCleanScript()
    check(_lib().sf_mlx_load(os.fsencode(path), C.byref(s)))
    return s


def clean_file(in_ply, out_ply, script_mlx):
    """meshlabserver -i in_ply -o out_ply -m vc -s script_mlx"""
    s = load_script(script_mlx)
    m = Mesh.read(in_ply)
    h, st = C.c_void_p(), SfCleanStats()
    check(_lib().sf_mesh_clean_script(m._h, C.byref(s), C.byref(h), C.byref(st)))
    out = Mesh(h)
    out.write_ply(out_ply)
    out.close()
    m.close()
    res = {n: getattr(st, n) for n, _ in SfCleanStats._fields_}
    if s.simplify:
        res["simplify"] = {n: getattr(s.simplify_stats, n) for n, _ in SfSimplifyStats._fields_}
    return res
