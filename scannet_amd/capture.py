"""ScannerApp captures -- host-side mirror of the Occipital depth codec (ScannerApp/depth2pgm/uplinksimple_*.h) and of the
`convert` stage (Converter/main.cpp) over the C ABI."""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import check


class SfCaptureMeta(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_color_frames", "num_depth_frames", "num_imu", "color_width", "color_height",
                                          "depth_width", "depth_height")] + \
               [(n, C.c_float) for n in ("fx_color", "fy_color", "mx_color", "my_color", "fx_depth", "fy_depth", "mx_depth", "my_depth")] + \
               [("color_to_depth_extrinsics", C.c_float * 16), ("has_extrinsics", C.c_int32)]


class SfConvertStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("frames", "depth_frames_in_capture", "imu_frames", "imu_skipped", "depth_stream_bytes")] + \
               [("threads", C.c_uint32)]


COLOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64))


def _lib():
    L = _abi.lib()
    vp, u64 = C.c_void_p, C.c_uint64
    L.sf_occ_decode.argtypes = [vp, u64, u64, vp]
    L.sf_occ_encode_bound.argtypes = [u64]
    L.sf_occ_encode_bound.restype = u64
    L.sf_occ_encode.argtypes = [vp, u64, vp, u64, C.POINTER(u64)]
    L.sf_occ_shift2depth.argtypes = [C.c_uint16]
    L.sf_occ_shift2depth.restype = C.c_uint16
    L.sf_occ_shift2depth_buffer.argtypes = [vp, u64, C.c_int]
    L.sf_capture_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.sf_capture_close.argtypes = [vp]
    L.sf_capture_close.restype = None
    L.sf_capture_get_meta.argtypes = [vp, C.POINTER(SfCaptureMeta)]
    L.sf_capture_decode_depth.argtypes = [vp, u64, vp, C.POINTER(u64)]
    L.sf_capture_convert.argtypes = [vp, C.c_char_p, vp, vp, C.c_int, C.c_int, C.POINTER(SfConvertStats)]
    return L


def decode(stream, num_elements):
    """uplinksimple::decode: bytes -> uint16 shift values."""
    buf = np.frombuffer(bytes(stream), np.uint8)
    out = np.empty(int(num_elements), np.uint16)
    check(_lib().sf_occ_decode(buf.ctypes.data if len(buf) else None, len(buf), int(num_elements), out.ctypes.data))
    return out


def encode(shift):
    """uplinksimple::encode: uint16 shift values (<= 2047) -> bytes."""
    a = np.ascontiguousarray(shift, np.uint16).ravel()
    L = _lib()
    cap = L.sf_occ_encode_bound(len(a))
    out = np.empty(cap, np.uint8)
    n = C.c_uint64(0)
    check(L.sf_occ_encode(a.ctypes.data if len(a) else None, len(a), out.ctypes.data, cap, C.byref(n)))
    return out[:n.value].tobytes()


def shift2depth(shift, zero_invalid=False):
    """uplinksimple::shift2depth on an array (a copy); zero_invalid as Converter/main.cpp:89-93."""
    a = np.array(shift, np.uint16, copy=True)
    flat = a.reshape(-1)
    check(_lib().sf_occ_shift2depth_buffer(flat.ctypes.data, flat.size, 1 if zero_invalid else 0))
    return a


class Capture:
    """<base>.txt / .depth / .imu of one ScannerApp capture."""

    def __init__(self, path):
        self._h = C.c_void_p()
        check(_lib().sf_capture_open(os.fsencode(path), C.byref(self._h)))
        m = SfCaptureMeta()
        check(_lib().sf_capture_get_meta(self._h, C.byref(m)))
        self.meta = m

    def close(self):
        if self._h:
            _lib().sf_capture_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def depth(self, frame):
        """(H x W uint16 millimetres, time stamp in microseconds)"""
        out = np.empty((self.meta.depth_height, self.meta.depth_width), np.uint16)
        ts = C.c_uint64(0)
        check(_lib().sf_capture_decode_depth(self._h, int(frame), out.ctypes.data, C.byref(ts)))
        return out, ts.value

    def convert(self, out_sens, color_blobs=None, color_compression=2, threads=0):
        """Write the .sens; color_blobs: optional sequence of bytes per frame (JPEG blobs, or raw RGB with color_compression=0)."""
        st = SfConvertStats()
        keep = {}
        cb = None
        if color_blobs is not None:
            def fn(user, frame, blob, nbytes):
                b = color_blobs[frame]
                arr = (C.c_uint8 * len(b)).from_buffer_copy(b)
                keep["cur"] = arr
                blob[0] = C.cast(arr, C.POINTER(C.c_uint8))
                nbytes[0] = len(b)
                return 0
            cb = COLOR_FN(fn)
        check(_lib().sf_capture_convert(self._h, os.fsencode(out_sens), C.cast(cb, C.c_void_p) if cb else None, None, int(color_compression),
                                        int(threads), C.byref(st)))
        return {n: getattr(st, n) for n, _ in SfConvertStats._fields_}


def write_capture(base, depth_shift_frames, timestamps_s, meta_lines, imu_records=()):
    """Test / tooling helper: writes <base>.txt/.depth/.imu the way ScannerApp does (ViewController+Sensor.mm:52-96,796-805,
    ViewController.mm:531-574): per frame u32 size + stream, then the depth and the colour time stamps as doubles."""
    with open(base + ".depth", "wb") as f:
        for fr in depth_shift_frames:
            s = encode(fr)
            f.write(np.uint32(len(s)).tobytes())
            f.write(s)
        ts = np.asarray(timestamps_s, np.float64)
        f.write(ts.tobytes())
        f.write(ts.tobytes())
    with open(base + ".txt", "wb") as f:
        for k, v in meta_lines:
            f.write(("%s = %s\r\n" % (k, v)).encode())
    with open(base + ".imu", "wb") as f:
        for rec in imu_records:
            f.write(np.asarray(rec, np.float64).tobytes())
