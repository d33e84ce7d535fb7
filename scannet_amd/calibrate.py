"""`calibrate` stage -- host-side mirror of Calibrate/src/calibration.h (class Calib, Calibration::calibrateScan's frame body)
over the C ABI; the image operations run on the GPU (scannet_amd/csrc/calibrate.hip)."""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import check


class SfCalibParams(C.Structure):
    _fields_ = [("color_width", C.c_uint32), ("color_height", C.c_uint32), ("depth_width", C.c_uint32), ("depth_height", C.c_uint32),
                ("color_intrinsic", C.c_float * 16), ("depth_intrinsic", C.c_float * 16), ("depth_extrinsic", C.c_float * 16),
                ("color_dist", C.c_float * 5), ("depth_dist", C.c_float * 5)]


class SfLut(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("zres", C.c_int32), ("max_dist", C.c_float), ("data", C.POINTER(C.c_float))]


def _lib():
    L = _abi.lib()
    vp = C.c_void_p
    L.sf_calib_params_load.argtypes = [C.c_char_p, C.POINTER(SfCalibParams)]
    L.sf_lut_load.argtypes = [C.c_char_p, C.POINTER(SfLut)]
    L.sf_lut_free.argtypes = [C.POINTER(SfLut)]
    L.sf_lut_free.restype = None
    L.sf_calibrator_create.argtypes = [C.POINTER(SfCalibParams), C.POINTER(SfLut), C.c_float, C.c_int, C.POINTER(vp)]
    L.sf_calibrator_destroy.argtypes = [vp]
    L.sf_calibrator_destroy.restype = None
    L.sf_calibrator_run.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.sf_calibrator_run_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.sf_calibrate_sens.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(SfCalibrateStats)]
    L.sf_jpeg_encode.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    return L


class SfCalibrateStats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("frames_with_colour", C.c_uint64), ("skipped_existing", C.c_int32), ("already_aligned", C.c_int32),
                ("seconds_total", C.c_double), ("threads", C.c_uint32)]


def calibrate_sens(in_sens, out_sens, params_txt, lut_path=None, device=0, threads=0):
    """Calibration::calibrateScan(inSens, outSens, params, table): the whole stage on one file."""
    st = SfCalibrateStats()
    check(_lib().sf_calibrate_sens(os.fsencode(in_sens), os.fsencode(out_sens), os.fsencode(params_txt),
                                   os.fsencode(lut_path) if lut_path else None, int(device), int(threads), C.byref(st)))
    return {n: getattr(st, n) for n, _ in SfCalibrateStats._fields_}


def jpeg_encode(rgb, quality=90, subsample=True):
    """Baseline JPEG blob of an [H, W, 3] uint8 image (what the stage writes for TYPE_JPEG colour)."""
    a = np.ascontiguousarray(rgb, np.uint8)
    h, w = a.shape[:2]
    out = np.empty(a.size + 65536, np.uint8)
    n = C.c_uint64(0)
    check(_lib().sf_jpeg_encode(a.ctypes.data, w, h, int(quality), 1 if subsample else 0, out.ctypes.data, out.size, C.byref(n)))
    return out[:n.value].tobytes()


def jpeg_decode(blob, width, height, device=None, device_huffman=False):
    """Baseline JPEG -> [H, W, 3] uint8: on the host (device=None: what sf_sens_decode_color runs); entropy decoding on the host and
    reconstruction on GPU `device`; or (device_huffman) entropy decoding on the GPU too -- what sf_fuse_run does with a colour frame, the host
    only parses the headers.  The same bytes every way; device_huffman raises ScanfuseError (unsupported) for restart intervals, which
    sf_fuse_run entropy-decodes on the host."""
    L = _lib()
    b = np.frombuffer(blob, np.uint8)
    out = np.empty((height, width, 3), np.uint8)
    L.sf_jpeg_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sf_jpeg_decode_gpu.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.sf_jpeg_decode_gpu_huffman.argtypes = L.sf_jpeg_decode_gpu.argtypes
    if device is None:
        check(L.sf_jpeg_decode(b.ctypes.data, b.size, width, height, out.ctypes.data))
    elif device_huffman:
        check(L.sf_jpeg_decode_gpu_huffman(b.ctypes.data, b.size, width, height, int(device), out.ctypes.data))
    else:
        check(L.sf_jpeg_decode_gpu(b.ctypes.data, b.size, width, height, int(device), out.ctypes.data))
    return out


def write_params(path, p):
    """A parameter file with the keys Calib::readFromFile reads (calibration.h:22-48)."""
    with open(path, "w") as f:
        f.write("colorWidth = %d\ncolorHeight = %d\ndepthWidth = %d\ndepthHeight = %d\n" % (p.color_width, p.color_height, p.depth_width, p.depth_height))
        for tag, K, dist in (("color", p.color_intrinsic, p.color_dist), ("depth", p.depth_intrinsic, p.depth_dist)):
            f.write("fx_%s = %r\nfy_%s = %r\nmx_%s = %r\nmy_%s = %r\n" % (tag, float(K[0]), tag, float(K[5]), tag, float(K[2]), tag, float(K[6])))
            for i in range(5):
                f.write("k%d_%s = %r\n" % (i + 1, tag, float(dist[i])))
        f.write("depthToColorExtrinsics = %s\n" % " ".join(repr(float(v)) for v in p.depth_extrinsic))


def make_params(color_wh, depth_wh, color_K, depth_K, depth_to_color=None, color_dist=(0,) * 5, depth_dist=(0,) * 5):
    """color_K / depth_K = (fx, fy, mx, my)."""
    p = SfCalibParams()
    p.color_width, p.color_height = color_wh
    p.depth_width, p.depth_height = depth_wh
    for K, dst in ((color_K, p.color_intrinsic), (depth_K, p.depth_intrinsic)):
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[1, 1], m[0, 2], m[1, 2] = K
        for i, v in enumerate(m.ravel()):
            dst[i] = v
    e = np.eye(4, dtype=np.float32) if depth_to_color is None else np.asarray(depth_to_color, np.float32).reshape(4, 4)
    for i, v in enumerate(e.ravel()):
        p.depth_extrinsic[i] = v
    for i in range(5):
        p.color_dist[i] = color_dist[i]
        p.depth_dist[i] = depth_dist[i]
    return p


def load_params(path):
    p = SfCalibParams()
    check(_lib().sf_calib_params_load(os.fsencode(path), C.byref(p)))
    return p


def write_lut(path, grid, max_dist):
    """Grid3D::WriteFile layout: grid[z, y, x] float32."""
    g = np.ascontiguousarray(grid, np.float32)
    with open(path, "wb") as f:
        f.write(np.array([g.shape[2], g.shape[1], g.shape[0]], np.int32).tobytes())
        f.write(np.float32(max_dist).tobytes())
        f.write(g.tobytes())


class Calibrator:
    def __init__(self, params, lut_grid=None, lut_max_dist=0.0, lut_path=None, depth_shift=1000.0, device=0):
        self._h = C.c_void_p()
        self.params = params
        L = _lib()
        lut = None
        self._keep = None
        if lut_path is not None:
            lut = SfLut()
            check(L.sf_lut_load(os.fsencode(lut_path), C.byref(lut)))
        elif lut_grid is not None:
            g = np.ascontiguousarray(lut_grid, np.float32)
            self._keep = g
            lut = SfLut(g.shape[2], g.shape[1], g.shape[0], float(lut_max_dist), g.ctypes.data_as(C.POINTER(C.c_float)))
        try:
            check(L.sf_calibrator_create(C.byref(params), C.byref(lut) if lut is not None else None, float(depth_shift), int(device), C.byref(self._h)))
        finally:
            if lut_path is not None:
                L.sf_lut_free(C.byref(lut))

    def close(self):
        if self._h:
            _lib().sf_calibrator_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self, depth_frames, rgb_frames=None):
        """depth_frames: [n, H, W] uint16; rgb_frames: [n, Hc, Wc, 3] uint8 or None -> (depth_out, rgb_out or None)."""
        d = np.ascontiguousarray(depth_frames, np.uint16)
        n = d.shape[0]
        dout = np.empty_like(d)
        vp = C.c_void_p
        arr = lambda ptrs: (vp * n)(*ptrs)
        di = arr([d[i].ctypes.data for i in range(n)])
        do = arr([dout[i].ctypes.data for i in range(n)])
        ri = ro = None
        rout = None
        if rgb_frames is not None:
            r = np.ascontiguousarray(rgb_frames, np.uint8)
            rout = np.empty_like(r)
            ri = arr([r[i].ctypes.data for i in range(n)])
            ro = arr([rout[i].ctypes.data for i in range(n)])
        check(_lib().sf_calibrator_run(self._h, n, ri, ro, di, do))
        return dout, rout
