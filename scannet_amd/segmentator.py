"""Segmentator -- host-side mirror of Segmentator/segmentator.cpp over the C ABI.

`segment(mesh_file, kthr, seg_min_verts)` mirrors `segment()` (segmentator.cpp:123-251) and returns the
segIndices; `segment_to_json` mirrors `main` + `writeToJSON` (:253-289) including the output file naming.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import check


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Mesh:
    """Triangle mesh handle (PLY/OBJ in, marching cubes out, PLY out)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def read(cls, path):
        h = C.c_void_p()
        check(_abi.lib().sf_ply_read(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, xyz, tris, rgba=None, keys=None, face_keys=None):
        """keys (one u64 per vertex) [+ face_keys (one per face)]: a marching-cubes mesh as Mesh.arrays(keys=True) / Mesh.face_keys() gave it --
        what sf_mesh_merge_parts needs of another process's part."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        if rgba is not None:
            rgba = np.ascontiguousarray(rgba, np.uint8).reshape(-1, 4)
        L = _abi.lib()
        if keys is not None:
            keys = np.ascontiguousarray(keys, np.uint64).reshape(-1)
            if len(keys) != len(xyz):
                raise ValueError("one key per vertex")
            if face_keys is not None:
                face_keys = np.ascontiguousarray(face_keys, np.uint64).reshape(-1)
                if len(face_keys) != len(tris):
                    raise ValueError("one face key per face")
            L.sf_mesh_create_keyed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
            h = C.c_void_p()
            check(L.sf_mesh_create_keyed(_ptr(xyz), _ptr(rgba), _ptr(keys), len(xyz), _ptr(tris), _ptr(face_keys), len(tris), C.byref(h)))
            return cls(h)
        L.sf_mesh_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        check(L.sf_mesh_create(_ptr(xyz), _ptr(rgba), len(xyz), _ptr(tris), len(tris), C.byref(h)))
        return cls(h)

    def counts(self):
        nv, nf = C.c_uint64(0), C.c_uint64(0)
        check(_abi.lib().sf_mesh_counts(self._h, C.byref(nv), C.byref(nf)))
        return nv.value, nf.value

    def arrays(self, keys=False):
        nv, nf = self.counts()
        xyz = np.zeros((nv, 3), np.float32)
        rgba = np.zeros((nv, 4), np.uint8)
        tris = np.zeros((nf, 3), np.uint32)
        k = np.zeros(nv, np.uint64) if keys else None
        check(_abi.lib().sf_mesh_copy(self._h, _ptr(xyz), _ptr(rgba), _ptr(tris), _ptr(k)))
        return (xyz, rgba, tris, k) if keys else (xyz, rgba, tris)

    def face_keys(self):
        """Marching-cubes meshes: the cube key of every face (ascending)."""
        nv, nf = self.counts()
        k = np.zeros(nf, np.uint64)
        check(_abi.lib().sf_mesh_copy_face_keys(self._h, _ptr(k)))
        return k

    @classmethod
    def merge_parts(cls, parts):
        """The meshes the ranks of a partitioned scan extracted (Mesh handles with keys) -> the mesh one fuser would have extracted
        (sf_mesh_merge_parts; partition.merge_slab_meshes is the same rule on numpy arrays)."""
        L = _abi.lib()
        arr = (C.c_void_p * max(len(parts), 1))(*[p._h for p in parts])
        L.sf_mesh_merge_parts.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        check(L.sf_mesh_merge_parts(arr, len(parts), C.byref(h)))
        return cls(h)

    def write_ply(self, path):
        check(_abi.lib().sf_mesh_write_ply(self._h, os.fsencode(path)))

    def close(self):
        if getattr(self, "_h", None):
            _abi.lib().sf_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def segment_arrays(xyz, tris, kthr=0.01, seg_min_verts=20, device=None):
    """device: None = the host path (sf_segment_mesh); an int = vertex normals and edge weights on that GPU (sf_segment_mesh_gpu: the same labels)."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
    out = np.zeros(len(xyz), np.int32)
    if device is None:
        check(_abi.lib().sf_segment_mesh(_ptr(xyz), len(xyz), _ptr(tris), len(tris), float(kthr), int(seg_min_verts), _ptr(out)))
    else:
        L = _abi.lib()
        L.sf_segment_mesh_gpu.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_float, C.c_int, C.c_int, C.c_void_p]
        check(L.sf_segment_mesh_gpu(_ptr(xyz), len(xyz), _ptr(tris), len(tris), float(kthr), int(seg_min_verts), int(device), _ptr(out)))
    return out


def segment(mesh_file, kthr=0.01, seg_min_verts=20):
    m = Mesh.read(mesh_file)
    xyz, _, tris = m.arrays()
    m.close()
    return segment_arrays(xyz, tris, kthr, seg_min_verts)


def segment_to_json(mesh_file, kthr=0.01, seg_min_verts=20, out_json=None):
    """Writes <mesh minus ext>.<kThresh %f>.segs.json (or out_json); returns the number of segments."""
    n = C.c_uint64(0)
    check(_abi.lib().sf_segment_file(os.fsencode(mesh_file), float(kthr), int(seg_min_verts),
                                     None if out_json is None else os.fsencode(out_json), C.byref(n)))
    return n.value
