"""`.sens` container -- host-side mirror of the reference readers over the C ABI.

Attribute and method names follow SensReader/python/SensorData.py (classes SensorData / RGBDFrame: `sensor_name`,
`intrinsic_depth`, `frames[i].camera_to_world`, `decompress_depth`, `export_poses`, ...), semantics follow the
C++ codec SensReader/c++/src/sensorData.h.  Parsing, inflate and JPEG decode run in libscanfuse.so
(scannet_amd/csrc/sens.cpp, zlib_codec.cpp, jpeg.cpp); frames are lazy views into the memory-mapped file.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import check

COMPRESSION_TYPE_COLOR = {-1: 'unknown', 0: 'raw', 1: 'png', 2: 'jpeg'}
COMPRESSION_TYPE_DEPTH = {-1: 'unknown', 0: 'raw_ushort', 1: 'zlib_ushort', 2: 'occi_ushort'}


class SfSensInfo(C.Structure):
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


class SfSensFrameMeta(C.Structure):
    _fields_ = [("timestamp_color", C.c_uint64), ("timestamp_depth", C.c_uint64), ("color_bytes", C.c_uint64), ("depth_bytes", C.c_uint64)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def zlib_inflate(data, out_bytes, device=None):
    """Inflate a zlib stream into a bytes object of at most out_bytes (no Adler-32 check, like the reference).  device=k: on GPU k, the way
    sf_fuse_run inflates depth frames (csrc/inflate_gpu.hip) -- the stream must be ONE final fixed-Huffman block (what the reference's writer and
    this library's emit) that inflates to exactly out_bytes, a multiple of 4; ScanfuseError (unsupported) otherwise."""
    src = np.frombuffer(data, np.uint8)
    dst = np.empty(out_bytes, np.uint8)
    if device is not None:
        L = _abi.lib()
        L.sf_zlib_inflate_gpu.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        check(L.sf_zlib_inflate_gpu(_ptr(src), len(src), out_bytes, int(device), _ptr(dst)))
        return dst.tobytes()
    n = C.c_uint64(0)
    check(_abi.lib().sf_zlib_inflate(_ptr(src), len(src), _ptr(dst), out_bytes, C.byref(n)))
    return dst[:n.value].tobytes()


def zlib_deflate(data):
    src = np.frombuffer(data, np.uint8)
    L = _abi.lib()
    cap = int(L.sf_zlib_deflate_bound(len(src)))
    dst = np.empty(cap, np.uint8)
    n = C.c_uint64(0)
    check(L.sf_zlib_deflate(_ptr(src), len(src), _ptr(dst), cap, C.byref(n)))
    return dst[:n.value].tobytes()


class RGBDFrame:
    """One frame of a SensorData (lazy: blobs stay in the mapped file until decompress_* is called)."""

    def __init__(self, owner, index):
        self._o, self._i = owner, index
        pose = np.zeros(16, np.float32)
        valid = C.c_int(0)
        check(_abi.lib().sf_sens_pose(owner._h, index, _ptr(pose), C.byref(valid)))
        self.camera_to_world = pose.reshape(4, 4)
        self.valid_pose = bool(valid.value)  # False for the all -inf "tracking lost" pose
        m = SfSensFrameMeta()
        check(_abi.lib().sf_sens_frame_meta(owner._h, index, C.byref(m)))
        self.timestamp_color, self.timestamp_depth = m.timestamp_color, m.timestamp_depth
        self.color_size_bytes, self.depth_size_bytes = m.color_bytes, m.depth_bytes

    def _blobs(self):
        L = _abi.lib()
        L.sf_sens_frame_blobs.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        cp, dp, cn, dn = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        check(L.sf_sens_frame_blobs(self._o._h, self._i, C.byref(cp), C.byref(cn), C.byref(dp), C.byref(dn)))
        return (C.string_at(cp, cn.value) if cn.value else b""), (C.string_at(dp, dn.value) if dn.value else b"")

    @property
    def color_compressed(self):
        """The colour blob as stored (RGBDFrame::getColorCompressed, sensorData.h:418)."""
        return self._blobs()[0]

    @property
    def depth_compressed(self):
        """The depth blob as stored (RGBDFrame::getDepthCompressed, sensorData.h:421)."""
        return self._blobs()[1]

    # the reference's attribute and method names for the same things (SensorData.py:19-45)
    color_data = color_compressed
    depth_data = depth_compressed

    def decompress_depth_zlib(self):
        """zlib.decompress(self.depth_data), SensorData.py:31-32: the frame's pixels as bytes (W*H little-endian u16)."""
        return self.decompress_depth().tobytes()

    def decompress_color_jpeg(self):
        """imageio.imread(self.color_data), SensorData.py:42-43: [H, W, 3] uint8."""
        return self.decompress_color()

    def decompress_depth(self, compression_type=None):
        """-> uint16 array [depth_height, depth_width] (the reference returns the raw bytes of the same data)."""
        o = self._o
        out = np.empty((o.depth_height, o.depth_width), np.uint16)
        check(_abi.lib().sf_sens_decode_depth(o._h, self._i, _ptr(out)))
        return out

    def compute_depth_image(self):
        """SensorData::computeDepthImage (sensorData.h:968-982): float32 [depth_height, depth_width] in metres, 0 where there is no measurement."""
        o = self._o
        out = np.empty((o.depth_height, o.depth_width), np.float32)
        check(_abi.lib().sf_sens_depth_image(o._h, self._i, _ptr(out)))
        return out

    def decompress_color(self, compression_type=None):
        o = self._o
        out = np.empty((o.color_height, o.color_width, 3), np.uint8)
        check(_abi.lib().sf_sens_decode_color(o._h, self._i, _ptr(out)))
        return out


class SensorData:
    def __init__(self, filename=None, _handle=None):
        self.version = 4
        self._h = None
        if filename is not None:
            self.load(filename)
        elif _handle is not None:
            self._h = _handle
            self._refresh()

    # -- reading ---------------------------------------------------------------------------------------
    def load(self, filename):
        h = C.c_void_p()
        check(_abi.lib().sf_sens_open(os.fsencode(filename), C.byref(h)))
        self._h = h
        self._refresh()

    def _refresh(self):
        info = SfSensInfo()
        check(_abi.lib().sf_sens_get_info(self._h, C.byref(info)))
        self.sensor_name = info.sensor_name.decode("latin-1")
        self.intrinsic_color = np.array(info.color_intrinsic, np.float32).reshape(4, 4)
        self.extrinsic_color = np.array(info.color_extrinsic, np.float32).reshape(4, 4)
        self.intrinsic_depth = np.array(info.depth_intrinsic, np.float32).reshape(4, 4)
        self.extrinsic_depth = np.array(info.depth_extrinsic, np.float32).reshape(4, 4)
        self.color_compression_type = COMPRESSION_TYPE_COLOR.get(info.color_compression, 'unknown')
        self.depth_compression_type = COMPRESSION_TYPE_DEPTH.get(info.depth_compression, 'unknown')
        self.color_width, self.color_height = info.color_width, info.color_height
        self.depth_width, self.depth_height = info.depth_width, info.depth_height
        self.depth_shift = info.depth_shift
        self.num_frames, self.num_imu_frames = info.num_frames, info.num_imu
        self.version = info.version
        self._frames = None

    @property
    def frames(self):
        if self._frames is None:
            self._frames = [RGBDFrame(self, i) for i in range(self.num_frames)]
        return self._frames

    def close(self):
        if self._h:
            _abi.lib().sf_sens_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- writing (initDefault / addFrame / saveToFile, sensorData.h:891-921,1101-1109) ------------------
    @classmethod
    def create(cls, color_width, color_height, depth_width, depth_height, intrinsic_color, intrinsic_depth,
               color_compression=0, depth_compression=1, depth_shift=1000.0, sensor_name="Unknown",
               extrinsic_color=None, extrinsic_depth=None):
        info = SfSensInfo()
        info.version = 4
        info.color_width, info.color_height, info.depth_width, info.depth_height = color_width, color_height, depth_width, depth_height
        info.color_compression, info.depth_compression, info.depth_shift = color_compression, depth_compression, depth_shift
        eye = np.eye(4, dtype=np.float32)
        for name, m in (("color_intrinsic", intrinsic_color), ("color_extrinsic", eye if extrinsic_color is None else extrinsic_color),
                        ("depth_intrinsic", intrinsic_depth), ("depth_extrinsic", eye if extrinsic_depth is None else extrinsic_depth)):
            setattr(info, name, (C.c_float * 16)(*np.asarray(m, np.float32).reshape(16)))
        info.sensor_name = sensor_name.encode("latin-1")[:255]
        h = C.c_void_p()
        check(_abi.lib().sf_sens_create(C.byref(info), C.byref(h)))
        return cls(_handle=h)

    def save_to_images(self, output_folder, basename="frame-"):
        """SensorData::saveToImages (sensorData.h:1380-1466): _info.txt + frame-%06d.color.jpg|png / .depth.pgm / .pose.txt, the reference's bytes."""
        L = _abi.lib()
        L.sf_sens_save_to_images.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p]
        check(L.sf_sens_save_to_images(self._h, os.fsencode(output_folder), None if basename is None else basename.encode(), None, None))

    @classmethod
    def load_from_images(cls, folder, basename="frame-", color_ending=None):
        """SensorData::loadFromImages (sensorData.h:1468-1559): a folder as `bin/sens` / saveToImages writes it -> a SensorData in memory (then save())."""
        L = _abi.lib()
        L.sf_sens_load_from_images.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        check(L.sf_sens_load_from_images(os.fsencode(folder), None if basename is None else basename.encode(), None if color_ending is None else color_ending.encode(),
                                         C.byref(h)))
        return cls(_handle=h)

    def add_frame(self, depth, camera_to_world=None, color=None, timestamp_color=0, timestamp_depth=0):
        pose = np.ascontiguousarray(np.eye(4) if camera_to_world is None else camera_to_world, np.float32).reshape(16)
        d = None if depth is None else np.ascontiguousarray(depth, np.uint16)
        if d is not None and d.size != self.depth_width * self.depth_height:
            raise ValueError("depth frame size mismatch")
        c = None if color is None else np.ascontiguousarray(np.frombuffer(color, np.uint8) if isinstance(color, (bytes, bytearray)) else color, np.uint8)
        check(_abi.lib().sf_sens_add_frame(self._h, _ptr(c), 0 if c is None else c.size, _ptr(d), _ptr(pose), timestamp_color, timestamp_depth))
        self._refresh()

    def add_frame_blobs(self, depth_blob, camera_to_world=None, color_blob=None, timestamp_color=0, timestamp_depth=0):
        """A frame whose blobs are already compressed as the header says: stored as given."""
        pose = np.ascontiguousarray(np.eye(4) if camera_to_world is None else camera_to_world, np.float32).reshape(16)
        L = _abi.lib()
        L.sf_sens_add_frame_blobs.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
        cb, db = bytes(color_blob or b""), bytes(depth_blob or b"")
        check(L.sf_sens_add_frame_blobs(self._h, cb or None, len(cb), db or None, len(db), _ptr(pose), int(timestamp_color), int(timestamp_depth)))
        self._refresh()

    def save_point_cloud(self, ply_path, frame_from=0, frame_to=0):
        """SensorData::saveToPointCloud (sensorData.h:1564-1602): the valid depth pixels of frames [frame_from, frame_to) as coloured world-space points; returns the count."""
        L = _abi.lib()
        L.sf_sens_save_point_cloud.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        n = C.c_uint64(0)
        check(L.sf_sens_save_point_cloud(self._h, os.fsencode(ply_path), int(frame_from), int(frame_to), C.byref(n)))
        return n.value

    def add_depth_frames(self, depth, poses, timestamp0=0, timestamp_step=33333, threads=0):
        """n depth-only frames [n, H, W] uint16 with poses [n, 4, 4], compressed on `threads` threads (0 = all this process may use)."""
        d = np.ascontiguousarray(depth, np.uint16)
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        if d.ndim != 3 or d.shape[1] * d.shape[2] != self.depth_width * self.depth_height or len(p) != len(d):
            raise ValueError("depth must be [n, H, W] at the file's depth size with one pose per frame")
        L = _abi.lib()
        L.sf_sens_add_depth_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
        check(L.sf_sens_add_depth_frames(self._h, _ptr(d), self.depth_width * self.depth_height * 2, len(d), _ptr(p), int(timestamp0), int(timestamp_step), int(threads)))
        self._refresh()

    # -- IMU frames (sensorData.h:760-858): 128 bytes each = rotationRate, acceleration, magneticField, attitude, gravity (5 x 3 doubles) + time stamp (us)
    IMU_DTYPE = np.dtype([("rotationRate", "<f8", 3), ("acceleration", "<f8", 3), ("magneticField", "<f8", 3), ("attitude", "<f8", 3), ("gravity", "<f8", 3),
                          ("timeStamp", "<u8")])

    def add_imu_frame(self, record):
        """record: an IMU_DTYPE scalar (or anything numpy turns into one)."""
        r = np.asarray(record, self.IMU_DTYPE).reshape(1)
        check(_abi.lib().sf_sens_add_imu(self._h, _ptr(r)))
        self._refresh()

    @property
    def imu_frames(self):
        """m_IMUFrames as one structured array."""
        L = _abi.lib()
        L.sf_sens_imu.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        out = np.zeros(self.num_imu_frames, self.IMU_DTYPE)
        for i in range(self.num_imu_frames):
            check(L.sf_sens_imu(self._h, i, out[i:i + 1].ctypes.data))
        return out

    def find_closest_imu_frame(self, frame, based_on_rgb=True):
        """SensorData::findClosestIMUFrame(frameIdx, basedOnRGB) (sensorData.h:1000-1044): -> (index, IMU_DTYPE record)."""
        L = _abi.lib()
        L.sf_sens_find_closest_imu.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        out, idx = np.zeros(1, self.IMU_DTYPE), C.c_uint64(0)
        check(L.sf_sens_find_closest_imu(self._h, int(frame), 1 if based_on_rgb else 0, out.ctypes.data, C.byref(idx)))
        return idx.value, out[0]

    # -- editing in memory (then save): SensorData::replaceDepth / replaceColor / append / operator== (sensorData.h:948-964,1605-1650) ------------------
    def replace_depth(self, frame, depth):
        d = np.ascontiguousarray(depth, np.uint16)
        if d.shape != (self.depth_height, self.depth_width):
            raise ValueError("depth must be [depth_height, depth_width]")
        check(_abi.lib().sf_sens_replace_depth(self._h, int(frame), _ptr(d)))
        self._frames = None

    def replace_color(self, frame, color):
        """color: [color_height, color_width, 3] uint8 for a raw-colour file, an encoded JPEG / PNG blob (bytes) otherwise."""
        L = _abi.lib()
        L.sf_sens_replace_color.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        if isinstance(color, (bytes, bytearray)):
            b = np.frombuffer(bytes(color), np.uint8)
        else:
            b = np.ascontiguousarray(color, np.uint8).reshape(-1)
        check(L.sf_sens_replace_color(self._h, int(frame), _ptr(b), b.size))
        self._frames = None

    def append(self, second):
        check(_abi.lib().sf_sens_append(self._h, second._h))
        self._refresh()

    def apply_transform(self, t):
        """SensorData::applyTransform: every tracked pose <- t @ pose (4x4, row-major); "tracking lost" poses stay."""
        m = np.ascontiguousarray(t, np.float32).reshape(16)
        check(_abi.lib().sf_sens_apply_transform(self._h, _ptr(m)))
        self._frames = None

    def __eq__(self, other):
        if not isinstance(other, SensorData):
            return NotImplemented
        eq = C.c_int(0)
        check(_abi.lib().sf_sens_equal(self._h, other._h, C.byref(eq)))
        return bool(eq.value)

    __hash__ = None

    def set_pose(self, frame, camera_to_world):
        pose = np.ascontiguousarray(camera_to_world, np.float32).reshape(16)
        check(_abi.lib().sf_sens_set_pose(self._h, frame, _ptr(pose)))
        self._frames = None

    def save(self, filename):
        check(_abi.lib().sf_sens_save(self._h, os.fsencode(filename)))

    # -- exports: the file layout of SensorData.py:76-124 (one "%f %f %f %f" line per matrix row) ------------------------------------
    @staticmethod
    def save_mat_to_file(matrix, filename):
        rows = np.asarray(matrix, np.float64).reshape(-1, np.asarray(matrix).shape[-1])
        with open(filename, "w") as out:
            out.write("".join(" ".join("%f" % v for v in row) + "\n" for row in rows))

    def export_poses(self, output_path, frame_skip=1):
        """<output_path>/<frame index>.txt = camera-to-world of every frame_skip-th frame."""
        os.makedirs(output_path, exist_ok=True)
        for index, frame in list(enumerate(self.frames))[::frame_skip]:
            self.save_mat_to_file(frame.camera_to_world, os.path.join(output_path, "%d.txt" % index))

    def export_intrinsics(self, output_path):
        """intrinsic_color / extrinsic_color / intrinsic_depth / extrinsic_depth .txt"""
        os.makedirs(output_path, exist_ok=True)
        for kind in ("intrinsic", "extrinsic"):
            for camera in ("color", "depth"):
                name = "%s_%s" % (kind, camera)
                self.save_mat_to_file(getattr(self, name), os.path.join(output_path, name + ".txt"))

    @staticmethod
    def _for_frames(fn, indices):
        """fn(i) for every index on a small pool of threads: the library calls (inflate, PNG / JPEG coding) run outside the interpreter lock, so a scan's
        worth of exports takes the cores' time, not one core's; the first exception is raised after the pool has drained."""
        from concurrent.futures import ThreadPoolExecutor
        indices = list(indices)
        workers = max(1, min(8, os.cpu_count() or 1, len(indices)))
        if workers == 1:
            for i in indices:
                fn(i)
            return
        with ThreadPoolExecutor(workers) as pool:
            for _ in pool.map(fn, indices):
                pass

    @staticmethod
    def _resize_nearest(image, image_size):
        """cv2.resize(image, (image_size[1], image_size[0]), interpolation=cv2.INTER_NEAREST) as SensorData.py:84,99 calls it (image_size = (height,
        width)): destination pixel (y, x) takes source pixel (min(floor(y * H / h), H - 1), min(floor(x * W / w), W - 1))."""
        h, w = int(image_size[0]), int(image_size[1])
        H, W = image.shape[:2]
        ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
        xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
        return np.ascontiguousarray(image[ys][:, xs])

    def export_depth_images(self, output_path, image_size=None, frame_skip=1):
        """<output_path>/<frame index>.png: 16-bit grey PNG of every frame_skip-th depth frame (SensorData.py:78-91; written by sf_png_write, the
        reference goes through pypng), optionally resized to image_size = (height, width) with nearest-neighbour sampling."""
        os.makedirs(output_path, exist_ok=True)
        L = _abi.lib()
        L.sf_png_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        frames = self.frames

        def one(f):
            d = frames[f].decompress_depth()
            if image_size is not None:
                d = self._resize_nearest(d, image_size)
            check(L.sf_png_write(os.fsencode(os.path.join(output_path, str(f) + ".png")), _ptr(d), d.shape[1], d.shape[0], 1, 16))
        self._for_frames(one, range(0, len(frames), frame_skip))

    def export_color_images(self, output_path, image_size=None, frame_skip=1):
        """<output_path>/<frame index>.jpg of every frame_skip-th colour frame (SensorData.py:93-101).  The reference decodes every frame and encodes
        it again (imageio); a stored JPEG that is not resized is written as it is here -- the camera's own bytes, no second generation loss --,
        anything else (raw / PNG colour, or image_size given) is decoded, resized nearest-neighbour and encoded by sf_jpeg_encode (quality 75,
        imageio's default)."""
        from . import calibrate
        os.makedirs(output_path, exist_ok=True)
        frames = self.frames

        def one(f):
            frame = frames[f]
            if image_size is None and self.color_compression_type == "jpeg":
                blob = frame.color_compressed
            else:
                color = frame.decompress_color()
                if image_size is not None:
                    color = self._resize_nearest(color, image_size)
                blob = calibrate.jpeg_encode(color, 75, True)
            with open(os.path.join(output_path, str(f) + ".jpg"), "wb") as out:
                out.write(blob)
        self._for_frames(one, range(0, len(frames), frame_skip))


class SensorDataWriter:
    """Frames streamed into a .sens as they arrive (SensorData::LiveSensorDataWriter, sensorData.h:1112-1246): `cache_frames` frames of memory instead of a
    whole scan's.  The file is what SensorData.create + add_frame + save write.

        with SensorDataWriter(path, color_width, ..., sensor_name="StructureSensor") as w:
            w.add_frame(depth, pose, color=jpeg_bytes)
        print(w.frames_written, w.path)          # path: the name really used (overwrite=False counts a numeric suffix up past existing files)
    """

    def __init__(self, filename, color_width, color_height, depth_width, depth_height, intrinsic_color, intrinsic_depth, extrinsic_color=None, extrinsic_depth=None,
                 color_compression=0, depth_compression=1, depth_shift=1000.0, sensor_name="StructureSensor", overwrite=True, cache_frames=0):
        info = SfSensInfo()
        info.version = 4
        info.color_width, info.color_height, info.depth_width, info.depth_height = color_width, color_height, depth_width, depth_height
        info.color_compression, info.depth_compression, info.depth_shift = color_compression, depth_compression, depth_shift
        eye = np.eye(4, dtype=np.float32)
        for name, m in (("color_intrinsic", intrinsic_color), ("depth_intrinsic", intrinsic_depth), ("color_extrinsic", eye if extrinsic_color is None else extrinsic_color),
                        ("depth_extrinsic", eye if extrinsic_depth is None else extrinsic_depth)):
            getattr(info, name)[:] = list(np.asarray(m, np.float32).reshape(16))
        info.sensor_name = sensor_name.encode("latin-1")
        L = _abi.lib()
        L.sf_sens_writer_open.argtypes = [C.POINTER(SfSensInfo), C.c_char_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sf_sens_writer_path.restype = C.c_char_p
        L.sf_sens_writer_path.argtypes = [C.c_void_p]
        L.sf_sens_writer_add_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.sf_sens_writer_add_frame_blobs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
        L.sf_sens_writer_close.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self._h = C.c_void_p()
        check(L.sf_sens_writer_open(C.byref(info), os.fsencode(filename), 1 if overwrite else 0, int(cache_frames), C.byref(self._h)))
        self.path = os.fsdecode(L.sf_sens_writer_path(self._h))
        self.frames_written = 0
        self._shape = (depth_height, depth_width)

    @staticmethod
    def _bytes(a):
        if a is None:
            return None, 0
        b = np.frombuffer(bytes(a), np.uint8) if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a, np.uint8).reshape(-1)
        return b, b.size

    def add_frame(self, depth, camera_to_world=None, color=None, timestamp_color=0, timestamp_depth=0):
        d = None if depth is None else np.ascontiguousarray(depth, np.uint16)
        if d is not None and d.shape != self._shape:
            raise ValueError("depth must be [depth_height, depth_width]")
        pose = np.ascontiguousarray(np.eye(4) if camera_to_world is None else camera_to_world, np.float32).reshape(16)
        c, cn = self._bytes(color)
        check(_abi.lib().sf_sens_writer_add_frame(self._h, _ptr(c), cn, _ptr(d), _ptr(pose), int(timestamp_color), int(timestamp_depth)))

    def add_frame_blobs(self, depth_blob, camera_to_world=None, color_blob=None, timestamp_color=0, timestamp_depth=0):
        pose = np.ascontiguousarray(np.eye(4) if camera_to_world is None else camera_to_world, np.float32).reshape(16)
        c, cn = self._bytes(color_blob)
        d, dn = self._bytes(depth_blob)
        check(_abi.lib().sf_sens_writer_add_frame_blobs(self._h, _ptr(c), cn, _ptr(d), dn, _ptr(pose), int(timestamp_color), int(timestamp_depth)))

    def close(self):
        if self._h:
            n = C.c_uint64(0)
            h, self._h = self._h, None
            rc = _abi.lib().sf_sens_writer_close(h, C.byref(n))
            self.frames_written = n.value
            check(rc)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

