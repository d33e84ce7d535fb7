"""Voxel-hash TSDF fuser -- host-side mirror of the scene-representation interface of the reference's
`improve` stage (external DepthSensing.exe, call site Server/scan_processor.py:137-138).

Thin ctypes layer over the C ABI (include/scanfuse.h); all arithmetic runs in the HIP kernels of
scannet_amd/csrc/fuser.hip.  There is no CPU fallback: creating a Fuser without an MI355X raises.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import SfParams, SfStats, check

class SfRunStats(C.Structure):
    _fields_ = [("frames_total", C.c_uint64), ("frames_integrated", C.c_uint64), ("frames_skipped", C.c_uint64),
                ("decode_threads", C.c_uint32), ("color_fused", C.c_uint32),
                ("seconds_total", C.c_double), ("seconds_decode_cpu", C.c_double)]


VOXEL_DTYPE = np.dtype([("sdf", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("w", "u1")])


def default_params(**over):
    """zParametersScanNet.txt values with BASELINE.json's 4 mm / 2^19-bucket overrides; keyword overrides."""
    p = SfParams()
    _abi.lib().sf_params_default(C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError("sf_params has no field %r" % k)
        setattr(p, k, v)
    return p


def load_params(path, base=None):
    """Parse an mLib ParameterFile (e.g. Server/tools/recons/zParametersScanNet.txt)."""
    p = base if base is not None else default_params()
    check(_abi.lib().sf_params_load_file(str(path).encode(), C.byref(p)))
    return p


def device_count():
    n = C.c_int(0)
    rc = _abi.lib().sf_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):  # torch tensor
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class Fuser:
    def __init__(self, params=None, device=0, **tune):
        self.params = params if params is not None else default_params()
        self.device = device
        h = C.c_void_p()
        check(_abi.lib().sf_fuser_create(C.byref(self.params), int(device), C.byref(h)))
        self._h = h
        self.tune(**tune)

    def tune(self, **switches):
        """Scheduling switches (include/scanfuse_internal.h sf_fuser_tune: batch, overlap, xcd_walk, pipe, pipe_wgs, alloc_group);
        the voxels are bit-identical under all of them -- bench.py and the tests use this, a pipeline stage never needs to."""
        L = _abi.lib()
        L.sf_fuser_tune.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        for k, v in switches.items():
            check(L.sf_fuser_tune(self._h, k.encode(), int(v)))
        return self

    def close(self):
        if getattr(self, "_h", None):
            _abi.lib().sf_fuser_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- host buffers -------------------------------------------------------------------------------
    def _host_frame(self, fn, depth, pose, rgb):
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        if depth.size != self.params.depth_width * self.params.depth_height:
            raise ValueError("depth frame has %d pixels, fuser expects %dx%d" % (depth.size, self.params.depth_width, self.params.depth_height))
        pose = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            want = (self.params.color_width * self.params.color_height if self.params.color_width > 0 else depth.size) * 3
            if rgb.size != want:
                raise ValueError("rgb must be HxWx3 at depth resolution (or at the colour resolution given in sf_params)")
        rc = check(fn(self._h, _ptr(depth), _ptr(rgb), _ptr(pose)), allow=(_abi.SF_ERR_SKIPPED,))
        return rc == 0

    def integrate(self, depth, pose, rgb=None):
        """Fuse one frame; returns False when the frame was skipped (pose all -inf)."""
        return self._host_frame(_abi.lib().sf_fuser_integrate, depth, pose, rgb)

    def deintegrate(self, depth, pose, rgb=None):
        return self._host_frame(_abi.lib().sf_fuser_deintegrate, depth, pose, rgb)

    # -- device buffers (torch tensors or raw device pointers) --------------------------------------
    def integrate_device(self, d_depth, pose, d_rgb=None):
        pose = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        rc = check(_abi.lib().sf_fuser_integrate_device(self._h, _ptr(d_depth), _ptr(d_rgb), _ptr(pose)), allow=(_abi.SF_ERR_SKIPPED,))
        return rc == 0

    def deintegrate_device(self, d_depth, pose, d_rgb=None):
        pose = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        rc = check(_abi.lib().sf_fuser_deintegrate_device(self._h, _ptr(d_depth), _ptr(d_rgb), _ptr(pose)), allow=(_abi.SF_ERR_SKIPPED,))
        return rc == 0

    def integrate_batch_device(self, d_depth, frame_stride_bytes, poses, d_rgb=None, rgb_stride_bytes=0):
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
        if d_rgb is None:
            check(_abi.lib().sf_fuser_integrate_batch_device(self._h, _ptr(d_depth), int(frame_stride_bytes), _ptr(poses), len(poses)))
        else:
            L = _abi.lib()
            L.sf_fuser_integrate_batch_device_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
            check(L.sf_fuser_integrate_batch_device_rgb(self._h, _ptr(d_depth), int(frame_stride_bytes), _ptr(d_rgb), int(rgb_stride_bytes), _ptr(poses), len(poses)))

    @property
    def batch_frames(self):
        """Frames fused per pass over the voxel tiles by integrate_batch_device / run (default 16; tune(batch=...))."""
        return int(_abi.lib().sf_fuser_batch_frames(self._h))

    def reset(self):
        """Empty volume again; parameters, streams and tuning stay (sf_fuser_reset)."""
        L = _abi.lib()
        L.sf_fuser_reset.argtypes = [C.c_void_p]
        check(L.sf_fuser_reset(self._h))

    def garbage_collect(self):
        n = C.c_uint32(0)
        check(_abi.lib().sf_fuser_garbage_collect(self._h, C.byref(n)))
        return n.value

    def sync(self):
        check(_abi.lib().sf_fuser_sync(self._h))

    @property
    def stream(self):
        return _abi.lib().sf_fuser_stream(self._h)

    def stats(self):
        s = SfStats()
        check(_abi.lib().sf_fuser_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in SfStats._fields_}

    def profile(self, on=True):
        check(_abi.lib().sf_fuser_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms, n, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        check(_abi.lib().sf_fuser_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def calib_tile_rmw(self, read_only=False, iters=20):
        """(average microseconds, tiles): the most recent pass's tile traffic without its arithmetic (pattern ceiling)."""
        us, n = C.c_double(0), C.c_uint32(0)
        check(_abi.lib().sf_fuser_calib_tile_rmw(self._h, 1 if read_only else 0, int(iters), C.byref(us), C.byref(n)))
        return us.value, n.value

    def calib_tile_rmw_ex(self, read_only=False, contiguous=False, tiles_per_turnaround=1, iters=10):
        """(average microseconds, tiles) of the most recent pass's tile traffic taken apart: scattered list or one contiguous span, G tiles read before G written."""
        L = _abi.lib()
        L.sf_fuser_calib_tile_rmw_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        mode = (1 if read_only else 0) | (2 if contiguous else 0) | ({1: 0, 2: 1, 4: 2}[int(tiles_per_turnaround)] << 2)
        us, n = C.c_double(0), C.c_uint32(0)
        check(L.sf_fuser_calib_tile_rmw_ex(self._h, mode, int(iters), C.byref(us), C.byref(n)))
        return us.value, n.value

    @staticmethod
    def prepare_run(sensor_data, params, device=0):
        """sf_fuse_run_prepare: with the file open and BEFORE the Fuser is created, start making the streams and rings a run of this file wants."""
        L = _abi.lib()
        L.sf_fuse_run_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        check(L.sf_fuse_run_prepare(sensor_data._h, C.byref(params), int(device)))

    def run(self, sensor_data, first=0, last=0, decode_threads=0):
        """Fuse frames [first, last) of a scannet_amd.sens.SensorData (threaded decode overlapped with the GPU)."""
        st = SfRunStats()
        check(_abi.lib().sf_fuse_run(self._h, sensor_data._h, int(first), int(last), int(decode_threads), C.byref(st)))
        out = {k: getattr(st, k) for k, _ in SfRunStats._fields_}
        L = _abi.lib()
        if hasattr(L, "sf_fuse_run_device_counts"):   # scanfuse_internal.h: where the frames were decoded
            c = (C.c_uint64 * 4)()
            L.sf_fuse_run_device_counts.argtypes = [C.POINTER(C.c_uint64)]
            check(L.sf_fuse_run_device_counts(c))
            out.update(depth_inflated_on_device=c[0], depth_inflated_on_host=c[1], jpeg_entropy_on_device=c[2], jpeg_entropy_on_host=c[3])
        return out

    # -- one large scan over several GPUs (scannet_amd/partition.py) -------------------------------------
    def set_slab(self, axis, lo_block, hi_block):
        """Only allocate blocks with lo_block <= coord[axis] < hi_block (axis < 0: no partition)."""
        check(_abi.lib().sf_fuser_set_slab(self._h, int(axis), int(lo_block), int(hi_block)))

    def export_blocks_where(self, axis, lo, hi, include_ghosts=False):
        """Live blocks with lo <= coord[axis] < hi -> (coords int32 [n,3], voxels VOXEL_DTYPE [n,512]), sorted by (x,y,z)."""
        L = _abi.lib()
        n = C.c_uint64(0)
        check(L.sf_fuser_export_blocks_where(self._h, int(axis), int(lo), int(hi), int(bool(include_ghosts)), None, None, 0, C.byref(n), 0))
        coords = np.zeros((n.value, 3), np.int32)
        vox = np.zeros((n.value, 512), VOXEL_DTYPE)
        if n.value:
            check(L.sf_fuser_export_blocks_where(self._h, int(axis), int(lo), int(hi), int(bool(include_ghosts)), _ptr(coords), _ptr(vox), n.value, C.byref(n), 0))
        order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
        return coords[order], vox[order]

    def import_blocks(self, coords, voxels, ghost=True):
        coords = np.ascontiguousarray(coords, np.int32).reshape(-1, 3)
        voxels = np.ascontiguousarray(voxels)
        if voxels.nbytes != len(coords) * 4096:
            raise ValueError("voxels must hold 4096 bytes per block")
        check(_abi.lib().sf_fuser_import_blocks(self._h, _ptr(coords), _ptr(voxels), len(coords), int(bool(ghost)), 0))

    def set_stripes(self, axis, origin_block, thickness_blocks, world, rank):
        """Own the stripes floor((coord[axis] - origin) / thickness) mod world == rank (sf_fuser_set_stripes)."""
        check(_abi.lib().sf_fuser_set_stripes(self._h, int(axis), int(origin_block), int(thickness_blocks), int(world), int(rank)))

    def count_boundary(self):
        n = C.c_uint64(0)
        check(_abi.lib().sf_fuser_export_boundary(self._h, None, None, 0, C.byref(n), 0))
        return n.value

    def export_boundary(self, coords=None, voxels=None):
        """The lowest block layer of each of this fuser's slabs / stripes.  Without arguments: numpy (coords int32 [n,3], voxels
        VOXEL_DTYPE [n,512]).  With `coords` / `voxels` = device buffers (torch CUDA tensors or raw pointers, at least count_boundary()
        blocks): written in place on the GPU, returns n -- nothing touches host memory."""
        L = _abi.lib()
        n = C.c_uint64(0)
        if coords is None:
            m = self.count_boundary()
            c = np.zeros((m, 3), np.int32)
            v = np.zeros((m, 512), VOXEL_DTYPE)
            if m:
                check(L.sf_fuser_export_boundary(self._h, _ptr(c), _ptr(v), m, C.byref(n), 0))
            return c, v
        cap = coords.shape[0] if hasattr(coords, "shape") else self.count_boundary()
        check(L.sf_fuser_export_boundary(self._h, _ptr(coords), _ptr(voxels), int(cap), C.byref(n), 1))
        return n.value

    def import_ghosts(self, coords, voxels, n=None):
        """Of a (gathered) payload keep the blocks this fuser needs as ghosts.  numpy arrays or device buffers; returns how many were kept."""
        L = _abi.lib()
        on_dev = not isinstance(coords, np.ndarray)
        if not on_dev:
            coords = np.ascontiguousarray(coords, np.int32).reshape(-1, 3)
            voxels = np.ascontiguousarray(voxels)
        cnt = int(coords.shape[0] if n is None else n)
        got = C.c_uint64(0)
        check(L.sf_fuser_import_ghosts(self._h, _ptr(coords), _ptr(voxels), cnt, 1 if on_dev else 0, C.byref(got)))
        return got.value

    def mc_timing(self):
        """Phases of the most recent extract_mesh() in ms (scanfuse_internal.h sf_fuser_mc_timing) + block / triangle / vertex counts."""
        L = _abi.lib()
        L.sf_fuser_mc_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        t = (C.c_double * 12)()
        check(L.sf_fuser_mc_timing(self._h, t, 12))
        names = ("total", "live_list", "count_pass", "scan_emit_pass", "vertex_sort", "heads_scan_weld", "triangle_sort_gather", "downloads_device", "host_alloc_and_wait")
        out = {k: round(t[i], 3) for i, k in enumerate(names)}
        out.update(blocks=int(t[9]), triangles=int(t[10]), vertices=int(t[11]))
        return out

    def extract_mesh(self):
        """Marching cubes over all live blocks -> segmentator.Mesh (vertices in edge-key order, deterministic)."""
        from .segmentator import Mesh
        h = C.c_void_p()
        check(_abi.lib().sf_fuser_extract_mesh(self._h, C.byref(h)))
        return Mesh(h)

    def export_blocks(self):
        """-> (coords int32 [n,3], voxels VOXEL_DTYPE [n,512]) sorted lexicographically by (x,y,z)."""
        n = C.c_uint64(0)
        check(_abi.lib().sf_fuser_export_blocks(self._h, None, None, 0, C.byref(n)))
        coords = np.zeros((n.value, 3), np.int32)
        vox = np.zeros((n.value, 512), VOXEL_DTYPE)
        if n.value:
            check(_abi.lib().sf_fuser_export_blocks(self._h, _ptr(coords), _ptr(vox), n.value, C.byref(n)))
        order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
        return coords[order], vox[order]
