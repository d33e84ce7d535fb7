"""ctypes binding of libscanfuse.so (include/scanfuse.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCANFUSE_LIBRARY: another build of the same ABI (tools/sanitize.py points it at the ASan/UBSan build of the host code)
LIB_PATH = os.environ.get("SCANFUSE_LIBRARY") or os.path.join(_HERE, "libscanfuse.so")


class SfParams(C.Structure):
    _fields_ = [
        ("depth_width", C.c_int32), ("depth_height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("mx", C.c_float), ("my", C.c_float),
        ("depth_shift", C.c_float), ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("voxel_size", C.c_float), ("trunc_base", C.c_float), ("trunc_scale", C.c_float),
        ("max_integration_dist", C.c_float),
        ("weight_sample", C.c_int32), ("weight_max", C.c_int32),
        ("mc_thresh_factor", C.c_float),
        ("hash_num_buckets", C.c_uint32), ("hash_bucket_size", C.c_uint32), ("num_sdf_blocks", C.c_uint32),
        ("mc_max_triangles", C.c_uint32), ("gc_enabled", C.c_int32),
        ("color_width", C.c_int32), ("color_height", C.c_int32),
        ("cfx", C.c_float), ("cfy", C.c_float), ("cmx", C.c_float), ("cmy", C.c_float),
        ("integration_width", C.c_int32), ("integration_height", C.c_int32),
        ("frustum_mode", C.c_int32), ("colour_round", C.c_int32), ("colour_first", C.c_int32), ("weight_mode", C.c_int32),
        ("weight_wrap", C.c_int32),
    ]


class SfStats(C.Structure):
    _fields_ = [
        ("frames_integrated", C.c_uint64), ("frames_skipped", C.c_uint64),
        ("blocks_allocated", C.c_uint32), ("heap_free", C.c_uint32),
        ("last_frame_blocks", C.c_uint32), ("alloc_failures", C.c_uint32),
        ("total_frame_blocks", C.c_uint64),
        ("hash_slots_used", C.c_uint32), ("high_water", C.c_uint32),
        ("total_pass_tiles", C.c_uint64),
    ]


SF_OK = 0
SF_ERR_SKIPPED = -8


class ScanfuseError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("scanfuse error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libscanfuse.so.  torch (when installed) is imported first so that the process holds ONE HIP
    runtime: torch bundles its own libamdhip64 with the same soname."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("scannet_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    L.sf_last_error.restype = C.c_char_p
    L.sf_version.restype = C.c_char_p
    L.sf_params_default.argtypes = [C.POINTER(SfParams)]
    L.sf_params_default.restype = None
    L.sf_params_load_file.argtypes = [C.c_char_p, C.POINTER(SfParams)]
    L.sf_device_count.argtypes = [C.POINTER(C.c_int)]
    L.sf_fuser_create.argtypes = [C.POINTER(SfParams), C.c_int, C.POINTER(vp)]
    L.sf_fuser_destroy.argtypes = [vp]
    L.sf_fuser_destroy.restype = None
    for name in ("sf_fuser_integrate", "sf_fuser_deintegrate", "sf_fuser_integrate_device", "sf_fuser_deintegrate_device"):
        getattr(L, name).argtypes = [vp, vp, vp, vp]
    L.sf_fuser_integrate_batch_device.argtypes = [vp, vp, u64, vp, u64]
    L.sf_fuser_garbage_collect.argtypes = [vp, C.POINTER(u32)]
    L.sf_fuser_sync.argtypes = [vp]
    L.sf_fuser_stats.argtypes = [vp, C.POINTER(SfStats)]
    L.sf_fuser_stream.argtypes = [vp]
    L.sf_fuser_stream.restype = vp
    L.sf_fuser_profile_enable.argtypes = [vp, C.c_int]
    L.sf_fuser_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]
    L.sf_fuser_export_blocks.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.sf_device_malloc.argtypes = [C.c_int, u64, C.POINTER(vp)]
    L.sf_device_free.argtypes = [vp]
    L.sf_device_upload.argtypes = [vp, vp, u64]
    L.sf_device_download.argtypes = [vp, vp, u64]
    L.sf_synth_room_device.argtypes = [vp, u64, u64, u64, u64, C.c_int, C.c_int, C.c_int, vp]
    _declare_optional(L)
    _lib = L
    return L


def _declare_optional(L):
    """Entry points added after the first slice; declared when present so that partial builds still load."""
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    table = {
        "sf_zlib_inflate": ([vp, u64, vp, u64, C.POINTER(u64)], C.c_int),
        "sf_zlib_deflate": ([vp, u64, vp, u64, C.POINTER(u64)], C.c_int),
        "sf_zlib_deflate_bound": ([u64], u64),
        "sf_sens_open": ([C.c_char_p, C.POINTER(vp)], C.c_int),
        "sf_sens_close": ([vp], None),
        "sf_sens_get_info": ([vp, vp], C.c_int),
        "sf_sens_decode_depth": ([vp, u64, vp], C.c_int),
        "sf_sens_decode_color": ([vp, u64, vp], C.c_int),
        "sf_sens_pose": ([vp, u64, vp, C.POINTER(C.c_int)], C.c_int),
        "sf_sens_frame_meta": ([vp, u64, vp], C.c_int),
        "sf_sens_create": ([vp, C.POINTER(vp)], C.c_int),
        "sf_sens_add_frame": ([vp, vp, u64, vp, vp, u64, u64], C.c_int),
        "sf_sens_set_pose": ([vp, u64, vp], C.c_int),
        "sf_sens_save": ([vp, C.c_char_p], C.c_int),
        "sf_fuse_run": ([vp, vp, u64, u64, C.c_int, vp], C.c_int),
        "sf_fuser_batch_frames": ([vp], C.c_int),
        "sf_fuser_set_slab": ([vp, C.c_int, i32, i32], C.c_int),
        "sf_fuser_export_blocks_where": ([vp, C.c_int, i32, i32, C.c_int, vp, vp, u64, C.POINTER(u64), C.c_int], C.c_int),
        "sf_fuser_import_blocks": ([vp, vp, vp, u64, C.c_int, C.c_int], C.c_int),
        "sf_fuser_set_stripes": ([vp, C.c_int, i32, i32, C.c_int, C.c_int], C.c_int),
        "sf_fuser_export_boundary": ([vp, vp, vp, u64, C.POINTER(u64), C.c_int], C.c_int),
        "sf_fuser_import_ghosts": ([vp, vp, vp, u64, C.c_int, C.POINTER(u64)], C.c_int),
        "sf_fuser_tune": ([vp, C.c_char_p, C.c_int], C.c_int),
        "sf_mesh_copy_face_keys": ([vp, vp], C.c_int),
        "sf_calib_stream": ([C.c_int, u64, C.c_int], C.c_int),
        "sf_fuser_calib_tile_rmw": ([vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)], C.c_int),
        "sf_fuser_extract_mesh": ([vp, C.POINTER(vp)], C.c_int),
        "sf_mesh_counts": ([vp, C.POINTER(u64), C.POINTER(u64)], C.c_int),
        "sf_mesh_copy": ([vp, vp, vp, vp, vp], C.c_int),
        "sf_mesh_write_ply": ([vp, C.c_char_p], C.c_int),
        "sf_mesh_free": ([vp], None),
        "sf_segment_mesh": ([vp, u64, vp, u64, C.c_float, C.c_int, vp], C.c_int),
        "sf_segment_file": ([C.c_char_p, C.c_float, C.c_int, C.c_char_p, C.POINTER(u64)], C.c_int),
        "sf_ply_read": ([C.c_char_p, C.POINTER(vp)], C.c_int),
    }
    for name, (args, res) in table.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res


def check(rc, allow=()):
    if rc != SF_OK and rc not in allow:
        raise ScanfuseError(rc, lib().sf_last_error().decode("utf-8", "replace"))
    return rc


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a container can show 256 logical CPUs
    and be allowed the time of 16 -- /sys/fs/cgroup/cpu.max); what thread pools and CPU baselines size themselves from."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)
