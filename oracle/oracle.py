"""ctypes loader for the CPU checkers under oracle/ -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (scannet_amd/) never does.  See oracle/tsdf_oracle.c for what is and is not pinned
by the reference ("parity unpinned" for the TSDF part).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False):
    """Compile oracle/liboracle.so and (when /root/reference is present) oracle/_ref/*."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


class OrParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("mx", C.c_float), ("my", C.c_float),
        ("depth_shift", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("voxel_size", C.c_float), ("trunc_base", C.c_float), ("trunc_scale", C.c_float),
        ("max_integration_dist", C.c_float),
        ("weight_sample", C.c_int32), ("weight_max", C.c_int32),
        ("mc_thresh_factor", C.c_float),
        ("frustum_mode", C.c_int32), ("colour_round", C.c_int32), ("colour_first", C.c_int32), ("weight_mode", C.c_int32),
        ("weight_wrap", C.c_int32),
    ]


def default_params(width=640, height=480, voxel=0.004):
    """SURVEY.md section 8d camera + Server/tools/recons/zParametersScanNet.txt:34-35,47-53 values."""
    return OrParams(width, height, 577.87, 577.87, (width - 1) / 2.0, (height - 1) / 2.0, 1000.0, 0.1, 6.0,
                    voxel, 0.06, 0.02, 4.0, 1, 255, 10.0)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.or_create.restype = C.c_void_p
        L.or_create.argtypes = [C.POINTER(OrParams), C.c_int]
        L.or_destroy.argtypes = [C.c_void_p]
        for f in (L.or_integrate, L.or_deintegrate):
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_garbage_collect.restype = C.c_int64
        L.or_garbage_collect.argtypes = [C.c_void_p]
        L.or_num_blocks.restype = C.c_int64
        L.or_num_blocks.argtypes = [C.c_void_p]
        L.or_export.restype = C.c_int64
        L.or_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_depth_to_float.argtypes = [C.POINTER(OrParams), C.c_void_p, C.c_void_p]
        L.or_mc_extract.restype = C.c_void_p
        L.or_mc_extract.argtypes = [C.c_void_p]
        L.or_mc_num_verts.restype = C.c_int64
        L.or_mc_num_verts.argtypes = [C.c_void_p]
        L.or_mc_num_tris.restype = C.c_int64
        L.or_mc_num_tris.argtypes = [C.c_void_p]
        L.or_mc_copy.argtypes = [C.c_void_p] * 5
        L.or_mc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


VOXEL_DTYPE = np.dtype([("sdf", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("w", "u1")])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Volume:
    """CPU voxel-hash TSDF volume (the executable form of DESIGN.md section 3)."""

    def __init__(self, params, threads=1):
        self.params = params
        self._h = lib().or_create(C.byref(params), int(threads))

    def close(self):
        if self._h:
            lib().or_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _frame(self, fn, depth, rgb, pose):
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        pose = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert depth.size == self.params.width * self.params.height
        return int(fn(self._h, _ptr(depth), _ptr(rgb), _ptr(pose)))

    def integrate(self, depth, pose, rgb=None):
        return self._frame(lib().or_integrate, depth, rgb, pose)

    def deintegrate(self, depth, pose, rgb=None):
        return self._frame(lib().or_deintegrate, depth, rgb, pose)

    def garbage_collect(self):
        return int(lib().or_garbage_collect(self._h))

    @property
    def num_blocks(self):
        return int(lib().or_num_blocks(self._h))

    def export(self):
        """-> (coords int32 [n,3], voxels VOXEL_DTYPE [n,512]) sorted lexicographically by (x,y,z)."""
        n = self.num_blocks
        coords = np.zeros((n, 3), np.int32)
        vox = np.zeros((n, 512), VOXEL_DTYPE)
        lib().or_export(self._h, _ptr(coords), _ptr(vox))
        return sort_blocks(coords, vox)

    def extract_mesh(self):
        """-> dict(pos f32[nv,3], col u8[nv,3], idx i32[nt,3], keys u64[nv]) in canonical order."""
        L = lib()
        m = L.or_mc_extract(self._h)
        nv, nt = int(L.or_mc_num_verts(m)), int(L.or_mc_num_tris(m))
        pos = np.zeros((nv, 3), np.float32)
        col = np.zeros((nv, 3), np.uint8)
        idx = np.zeros((nt, 3), np.int32)
        keys = np.zeros(nv, np.uint64)
        L.or_mc_copy(m, _ptr(pos), _ptr(col), _ptr(idx), _ptr(keys))
        L.or_mc_free(m)
        return dict(pos=pos, col=col, idx=idx, keys=keys)


def sort_blocks(coords, vox):
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    return coords[order], vox[order]


def depth_to_float(params, depth):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    out = np.zeros(depth.shape, np.float32)
    lib().or_depth_to_float(C.byref(params), _ptr(depth), _ptr(out))
    return out


# ---------------------------------------------------------------- compiled REFERENCE (oracle/_ref)
class RefSensInfo(C.Structure):
    _fields_ = [
        ("version", C.c_uint32),
        ("color_width", C.c_uint32), ("color_height", C.c_uint32), ("depth_width", C.c_uint32), ("depth_height", C.c_uint32),
        ("color_compression", C.c_int32), ("depth_compression", C.c_int32),
        ("depth_shift", C.c_float),
        ("num_frames", C.c_uint64), ("num_imu", C.c_uint64),
        ("color_intrinsic", C.c_float * 16), ("color_extrinsic", C.c_float * 16),
        ("depth_intrinsic", C.c_float * 16), ("depth_extrinsic", C.c_float * 16),
        ("sensor_name", C.c_char * 256),
    ]


_ref = None


def ref_sens_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_sens.so"))


def ref_unproject_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_unproject.so"))


def ref_unproject(K, depth_m):
    """The reference's own unprojection (filter.cu:74-91 on the float4x4 of cuda_SimpleMatrixUtil.h, compiled from /root/reference into
    oracle/_ref/libref_unproject.so): K 4x4 intrinsics, depth_m [H, W] float32 metres (-inf invalid) -> camera-space points [H, W, 3]."""
    L = C.CDLL(os.path.join(_HERE, "_ref", "libref_unproject.so"))
    L.ref_unproject.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
    K = np.ascontiguousarray(K, np.float32).reshape(16)
    d = np.ascontiguousarray(depth_m, np.float32)
    out = np.zeros(d.shape + (3,), np.float32)
    L.ref_unproject(_ptr(K), d.shape[1], d.shape[0], _ptr(d), _ptr(out))
    return out


def ref_intrinsics_inverse(K):
    L = C.CDLL(os.path.join(_HERE, "_ref", "libref_unproject.so"))
    L.ref_intrinsics_inverse.argtypes = [C.c_void_p, C.c_void_p]
    K = np.ascontiguousarray(K, np.float32).reshape(16)
    out = np.zeros(16, np.float32)
    L.ref_intrinsics_inverse(_ptr(K), _ptr(out))
    return out.reshape(4, 4)


def ref_segmentator_path(o0=False):
    p = os.path.join(_HERE, "_ref", "segmentator_ref_O0" if o0 else "segmentator_ref")
    return p if os.path.exists(p) else None


def ref_sens():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_sens.so"))
        L.ref_sens_open.restype = C.c_void_p
        L.ref_sens_open.argtypes = [C.c_char_p]
        L.ref_sens_close.argtypes = [C.c_void_p]
        L.ref_sens_get_info.argtypes = [C.c_void_p, C.POINTER(RefSensInfo)]
        L.ref_sens_decode_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.ref_sens_decode_color.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.ref_sens_pose.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.ref_sens_frame_meta.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_uint64)] * 4
        L.ref_sens_create.restype = C.c_void_p
        L.ref_sens_create.argtypes = [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p, C.c_float, C.c_char_p]
        L.ref_sens_add_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.ref_sens_save.argtypes = [C.c_void_p, C.c_char_p]
        if hasattr(L, "ref_sens_add_frames_mt"):
            L.ref_sens_add_frames_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
            L.ref_sens_depth_blob.restype = C.c_void_p
            L.ref_sens_depth_blob.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        if hasattr(L, "ref_sens_find_closest_imu"):
            L.ref_sens_add_imu.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_sens_find_closest_imu.restype = C.c_int64
            L.ref_sens_find_closest_imu.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        if hasattr(L, "ref_sens_equal"):
            L.ref_sens_append.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_sens_equal.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_sens_replace_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        _ref = L
    return _ref


def ref_write_sens(path, depth, poses, intrinsic_depth, rgb=None, intrinsic_color=None, threads=0, depth_shift=1000.0, name=b"StructureSensor",
                   timestamp_step=33333, want_blobs=False):
    """A .sens written by the REFERENCE writer: SensorData::initDefault + createFrame per frame (stb deflate, quality 8: sensorData.h:659-670 ->
    stb_image_write.h:721-823) + saveToFile, through oracle/_ref/libref_sens.so.  depth [n, H, W] uint16, poses [n, 4, 4]; rgb None (a file
    without colour: colour size 0 x 0, TYPE_RAW) or [n, Hc, Wc, 3] uint8 stored raw -- the reference cannot encode JPEG off Windows
    (sensorData.h:576-593).  Frames are compressed on `threads` threads (0 = all this process may use).  Returns the depth blobs when asked."""
    import numpy as np
    R = ref_sens()
    d = np.ascontiguousarray(depth, np.uint16)
    n, H, W = d.shape
    p = np.ascontiguousarray(poses, np.float32).reshape(n, 16)
    Kd = np.ascontiguousarray(intrinsic_depth, np.float32)
    Kc = Kd if intrinsic_color is None else np.ascontiguousarray(intrinsic_color, np.float32)
    c = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
    cw, ch = (0, 0) if c is None else (c.shape[2], c.shape[1])
    h = R.ref_sens_create(cw, ch, W, H, Kc.ctypes.data_as(C.c_void_p), Kd.ctypes.data_as(C.c_void_p), float(depth_shift), name)
    try:
        nt = threads if threads > 0 else len(os.sched_getaffinity(0))
        if R.ref_sens_add_frames_mt(h, None if c is None else c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n, p.ctypes.data_as(C.c_void_p), 0, int(timestamp_step), nt) != 0:
            raise RuntimeError("reference writer failed")
        blobs = None
        if want_blobs:
            blobs = []
            for i in range(n):
                nb = C.c_uint64(0)
                q = R.ref_sens_depth_blob(h, i, C.byref(nb))
                blobs.append(C.string_at(q, nb.value))
        if path is not None and R.ref_sens_save(h, os.fsencode(path)) != 0:
            raise RuntimeError("reference saveToFile failed: %s" % path)
        return blobs
    finally:
        R.ref_sens_close(h)


# ---------------------------------------------------------------- calibrate stage (oracle/calib_oracle.c)
class OrCalib(C.Structure):
    _fields_ = [("color_width", C.c_uint32), ("color_height", C.c_uint32), ("depth_width", C.c_uint32), ("depth_height", C.c_uint32),
                ("color_intrinsic", C.c_float * 16), ("depth_intrinsic", C.c_float * 16), ("depth_extrinsic", C.c_float * 16),
                ("color_dist", C.c_float * 5), ("depth_dist", C.c_float * 5)]


class OrLut(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("zres", C.c_int32), ("max_dist", C.c_float), ("data", C.c_void_p)]


def calib_lib():
    L = lib()
    vp = C.c_void_p
    L.or_calib_undistort_rgb.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    L.or_calib_undistort_f32.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_float]
    L.or_calib_undistort_distance.argtypes = [vp, C.c_int, C.c_int, C.POINTER(OrLut), C.c_float]
    L.or_calib_depth_to_color.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(OrCalib), C.c_float]
    L.or_calib_depth_to_color_splat.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(OrCalib), C.c_float]
    L.or_calib_frame.argtypes = [C.POINTER(OrCalib), C.POINTER(OrLut), C.c_float, vp, vp, vp, vp]
    return L


def calib_from(params):
    """OrCalib from a scannet_amd.calibrate.SfCalibParams-like object (same field names)."""
    o = OrCalib()
    for name, _ in OrCalib._fields_:
        v = getattr(params, name)
        if hasattr(v, "__len__"):
            for i in range(len(v)):
                getattr(o, name)[i] = v[i]
        else:
            setattr(o, name, v)
    return o


def calib_frame(cb, depth, rgb=None, lut_grid=None, lut_max_dist=0.0, shift=1000.0):
    """One frame through calibrateScan's body -> (depth_out uint16 [H, W], rgb_out or None)."""
    L = calib_lib()
    d = np.ascontiguousarray(depth, np.uint16)
    dout = np.empty_like(d)
    lut = None
    if lut_grid is not None:
        g = np.ascontiguousarray(lut_grid, np.float32)
        lut = OrLut(g.shape[2], g.shape[1], g.shape[0], float(lut_max_dist), g.ctypes.data)
    rin = rout = None
    if rgb is not None:
        rin = np.ascontiguousarray(rgb, np.uint8)
        rout = np.empty_like(rin)
    L.or_calib_frame(C.byref(cb), C.byref(lut) if lut is not None else None, float(shift), rin.ctypes.data if rin is not None else None,
                     rout.ctypes.data if rout is not None else None, d.ctypes.data, dout.ctypes.data)
    return dout, rout


# ---------------------------------------------------------------- 2-D annotation filter (oracle/filter2d_oracle.c)
def f2d_lib():
    L = lib()
    vp = C.c_void_p
    L.or_exp64.restype = C.c_double
    L.or_exp64.argtypes = [C.c_double]
    L.or_f2d_bilateral.argtypes = [vp, vp, C.c_float, C.c_float, C.c_int, C.c_int]
    L.or_f2d_resample_float.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.or_f2d_resample_uchar.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.or_f2d_vote.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    L.or_f2d_to_label.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.or_f2d_frame.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    return L


def f2d_frame(depth, rgb, instance, to_idx, to_inst, to_label):
    """Filter2dAnnotations.cpp:326-397 for one frame -> (instance_out, label_out)."""
    L = f2d_lib()
    d = np.ascontiguousarray(depth, np.uint16)
    c = np.ascontiguousarray(rgb, np.uint8)
    i = np.ascontiguousarray(instance, np.uint8)
    a, b, t = np.ascontiguousarray(to_idx, np.uint8), np.ascontiguousarray(to_inst, np.uint8), np.ascontiguousarray(to_label, np.uint16)
    ch, cw = c.shape[:2]
    io = np.empty((ch, cw), np.uint8)
    lo = np.empty((ch, cw), np.uint16)
    L.or_f2d_frame(d.ctypes.data, d.shape[1], d.shape[0], c.ctypes.data, cw, ch, i.ctypes.data, a.ctypes.data, b.ctypes.data, t.ctypes.data,
                   io.ctypes.data, lo.ctypes.data)
    return io, lo


# ---------------------------------------------------------------- annotation projection (oracle/project_oracle.c)
class OrProjectParams(C.Structure):
    _fields_ = [("color_width", C.c_uint32), ("color_height", C.c_uint32), ("depth_width", C.c_uint32), ("depth_height", C.c_uint32),
                ("fx", C.c_float), ("fy", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float), ("depth_dist_thresh", C.c_float),
                ("filter_using_original_depth", C.c_int32)]


def project_frame(params, xyz, tris, vinst, vlabel, cam2world, orig_depth=None, want_depth=False):
    """Visualizer.cpp:57-193 for one frame -> (instance, label[, rendered depth in metres])."""
    L = lib()
    vp = C.c_void_p
    L.or_project_frame.argtypes = [C.POINTER(OrProjectParams), vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    P = OrProjectParams(*[getattr(params, f) for f, _ in OrProjectParams._fields_])
    x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
    vi, vl = np.ascontiguousarray(vinst, np.uint8), np.ascontiguousarray(vlabel, np.uint16)
    c = np.ascontiguousarray(cam2world, np.float32).reshape(16)
    d = None if orig_depth is None else np.ascontiguousarray(orig_depth, np.uint16)
    io = np.empty((P.color_height, P.color_width), np.uint8)
    lo = np.empty((P.color_height, P.color_width), np.uint16)
    z = np.empty((P.color_height, P.color_width), np.float32) if want_depth else None
    rc = L.or_project_frame(C.byref(P), x.ctypes.data, len(x), t.ctypes.data, len(t), vi.ctypes.data, vl.ctypes.data, c.ctypes.data,
                            d.ctypes.data if d is not None else None, io.ctypes.data, lo.ctypes.data, z.ctypes.data if z is not None else None)
    assert rc == 0
    return (io, lo, z) if want_depth else (io, lo)


def project_matrix(cam2world, fx, fy, W, H, n, f):
    L = lib()
    L.or_project_matrix.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
    c = np.ascontiguousarray(cam2world, np.float32).reshape(16)
    M = np.zeros(16, np.float32)
    L.or_project_matrix(c.ctypes.data, fx, fy, W, H, n, f, M.ctypes.data)
    return M.reshape(4, 4)


# ---------------------------------------------------------------- decimate stage: the quadric collapse as rounds of independent collapses (oracle/simplify_rounds_oracle.c)
def simplify_rounds(xyz, tris, rgba=None, target_perc=0.2, target_faces=0, quality_thr=0.3, boundary_weight=1.0, optimal_placement=True, planar_quadric=False,
                    auto_clean=True):
    """The sequential restatement of scannet_amd/csrc/simplify_gpu.hip's rule (defaults: simplify.mlx:3-16).  Returns (xyz [v,3] f32, rgba [v,4] u8 or None,
    tris [f,3] u32, {"rounds", "collapses"})."""
    import numpy as np
    L = lib()
    L.or_simplify_rounds.restype = C.c_int
    L.or_simplify_rounds.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_float, C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    v = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
    c = None if rgba is None else np.ascontiguousarray(rgba, np.uint8).reshape(-1, 4)
    vo = np.zeros((max(len(v), 1), 3), np.float32)
    to = np.zeros((max(len(t), 1), 3), np.uint32)
    co = None if c is None else np.zeros((max(len(v), 1), 4), np.uint8)
    nv, nf, rounds, coll = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
    rc = L.or_simplify_rounds(v.ctypes.data, len(v), t.ctypes.data, len(t), None if c is None else c.ctypes.data, float(target_perc), int(target_faces), float(quality_thr),
                              float(boundary_weight), int(bool(optimal_placement)), int(bool(planar_quadric)), int(bool(auto_clean)), vo.ctypes.data, to.ctypes.data,
                              None if co is None else co.ctypes.data, C.byref(nv), C.byref(nf), C.byref(rounds), C.byref(coll))
    if rc != 0:
        raise RuntimeError("or_simplify_rounds failed (%d)" % rc)
    return vo[:nv.value], (None if co is None else co[:nv.value]), to[:nf.value], {"rounds": rounds.value, "collapses": coll.value}
