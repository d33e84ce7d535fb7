"""Deterministic synthetic RGB-D streams -- numpy host-side input generator.

Not on the compute path: this only fabricates the *inputs* (u16 depth frames + camToWorld poses) that
tests, bench.py and tools/make_synth_sens.py feed to the fuser, because there is no network for the
real ScanNet .sens files.  Conventions follow the reference codec: depth in millimetres with
depthShift = 1000 (SensReader/c++/src/sensorData.h:895,968-977), row-major camToWorld with the
translation in the last column (sensorData.h:186-196), camera looks along +z with x right / y down
and no y flip (sensorData.h:1568-1579), pixel centres at integer coordinates.
"""
import numpy as np

FX = 577.87  # SURVEY.md section 8d camera


def intrinsics(width=640, height=480):
    """(fx, fy, mx, my) of the synthetic StructureSensor-like camera."""
    return FX * width / 640.0, FX * width / 640.0, (width - 1) / 2.0, (height - 1) / 2.0


def intrinsic_matrix(width=640, height=480):
    fx, fy, mx, my = intrinsics(width, height)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[0, 2], m[1, 2] = fx, fy, mx, my
    return m


def plane_frame(width=640, height=480, depth_mm=2000):
    """Config 1: constant-depth plane perpendicular to the optical axis."""
    return np.full((height, width), depth_mm, np.uint16)


def yaw_pose(x, y, z, yaw):
    """camToWorld for a camera at (x,y,z), heading `yaw` (rad, about world +z), world z up."""
    c, s = np.cos(yaw), np.sin(yaw)
    m = np.eye(4, dtype=np.float64)
    m[:3, 0] = (s, -c, 0.0)   # camera x (right)
    m[:3, 1] = (0.0, 0.0, -1.0)  # camera y (down)
    m[:3, 2] = (c, s, 0.0)    # camera z (forward)
    m[:3, 3] = (x, y, z)
    return m.astype(np.float32)


ROOM = (6.0, 4.0, 3.0)


def trajectory_pose(i, n_frames, room=ROOM, inset=1.0, height=1.5, corner=0.5):
    """Config 2: closed rounded-rectangle walk inside a box room, yaw following the path."""
    lx, ly = room[0] - 2 * inset, room[1] - 2 * inset
    per = 2 * (lx + ly)
    s = (i % n_frames) / float(n_frames) * per
    # corner positions along the perimeter and headings of the four legs
    legs = [(lx, 0.0), (ly, np.pi / 2), (lx, np.pi), (ly, 3 * np.pi / 2)]
    x, y = inset, inset
    acc = 0.0
    yaw = 0.0
    for k, (ln, hd) in enumerate(legs):
        if s < acc + ln or k == 3:
            t = s - acc
            x += np.cos(hd) * t
            y += np.sin(hd) * t
            # blend the heading linearly within `corner` metres of the leg ends
            nxt = hd + np.pi / 2
            prv = hd - np.pi / 2
            if t > ln - corner:
                yaw = hd + (nxt - hd) * 0.5 * (t - (ln - corner)) / corner
            elif t < corner:
                yaw = hd + (prv - hd) * 0.5 * (corner - t) / corner
            else:
                yaw = hd
            break
        x += np.cos(hd) * ln
        y += np.sin(hd) * ln
        acc += ln
    return yaw_pose(x, y, height, yaw)


def render_room_depth(pose, width=640, height=480, room=ROOM, noise_frame=None):
    """Analytic depth (u16 mm) of an axis-aligned box room [0,rx]x[0,ry]x[0,rz] seen from inside.

    noise_frame: if not None, adds the SURVEY 8d LCG noise (3 LSBs) seeded with frame*W*H + pixel.
    """
    fx, fy, mx, my = intrinsics(width, height)
    u = (np.arange(width, dtype=np.float64) - mx) / fx
    v = (np.arange(height, dtype=np.float64) - my) / fy
    dc = np.stack(np.broadcast_arrays(u[None, :], v[:, None], np.ones((height, width))), -1)  # cam dirs, z = 1
    R = pose[:3, :3].astype(np.float64)
    o = pose[:3, 3].astype(np.float64)
    dw = dc @ R.T
    t = np.full((height, width), np.inf)
    for ax in range(3):
        d = dw[..., ax]
        with np.errstate(divide="ignore", invalid="ignore"):
            t_hi = np.where(d > 0, (room[ax] - o[ax]) / d, np.inf)
            t_lo = np.where(d < 0, (0.0 - o[ax]) / d, np.inf)
        t = np.minimum(t, np.minimum(t_hi, t_lo))
    mm = np.rint(t * 1000.0)
    mm = np.where(np.isfinite(mm) & (mm < 65535), mm, 0).astype(np.int64)
    if noise_frame is not None:
        pix = np.arange(width * height, dtype=np.uint64).reshape(height, width)
        s = (np.uint64(noise_frame) * np.uint64(width * height) + pix) & np.uint64(0xFFFFFFFF)
        s = (s * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        mm = np.where(mm > 0, mm + ((s >> np.uint64(24)) & np.uint64(7)).astype(np.int64), 0)
    return mm.astype(np.uint16)


def room_stream(n_frames, total_frames=None, width=640, height=480, noise=False, start=0):
    """Yield (depth u16 [H,W], pose f32 [4,4]) for frames start..start+n_frames-1 of the config-2 walk."""
    total = total_frames or n_frames
    for i in range(start, start + n_frames):
        pose = trajectory_pose(i, total)
        yield render_room_depth(pose, width, height, noise_frame=i if noise else None), pose


# ---- the other synthetic workloads of SURVEY 8d (bench.py --config scans / partition) ---------------------------------------------------
def scan_spec(index):
    """Config 4: scan `index` of the 1513-scan rebuild -- room size +-20 % and 300..6000 frames from an LCG seeded with the index."""
    s = (int(index) * 2654435761 + 12345) & 0xFFFFFFFF
    out = []
    for _ in range(4):
        for _ in range(3):   # neighbouring seeds stay close for the first steps of an LCG: mix between draws
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            s ^= s >> 15
        out.append((s >> 8) / float(1 << 24))
    room = (ROOM[0] * (0.8 + 0.4 * out[0]), ROOM[1] * (0.8 + 0.4 * out[1]), ROOM[2] * (0.8 + 0.4 * out[2]))
    return room, 300 + int(out[3] * 5700)


CORRIDOR_ROOMS = 10
CORRIDOR_PITCH = 6.5   # metres between room corners along x: 6 m rooms, 0.5 m walls (wider than the truncation band)


def corridor_room(frame, total_frames):
    """Config 5: the 50 000-frame walk through a corridor of 10 rooms -- (room index, frame inside the room, frames per room)."""
    per = -(-int(total_frames) // CORRIDOR_ROOMS)
    return int(frame) // per, int(frame) % per, per


def render_scan_device(d_depth, stride, first, n, total, width, height, room=ROOM, origin=(0.0, 0.0, 0.0), noise=True):
    """Render frames into device memory (sf_synth_scan_device, scanfuse_internal.h); returns the n poses [n, 16] float32."""
    import ctypes as C
    from . import _abi
    L = _abi.lib()
    L.sf_synth_scan_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    poses = np.zeros((n, 16), np.float32)
    r = np.asarray(room, np.float64)
    o = np.asarray(origin, np.float64)
    _abi.check(L.sf_synth_scan_device(C.c_void_p(int(d_depth)), int(stride), int(first), int(n), int(total), int(width), int(height), 1 if noise else 0,
                                      r.ctypes.data, o.ctypes.data, poses.ctypes.data))
    return poses
