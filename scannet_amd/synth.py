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


def clutter_boxes(room=ROOM, seed=0):
    """The 48 axis-aligned boxes of the furnished scene (scene 1): -> (lo float32 [48,3], hi float32 [48,3]) in room coordinates.  The same
    integer LCG and the same draws, in the same order, as clutter_boxes() in csrc/synth.hip (tests/test_synth.py compares the two):
    28 floor-standing boxes flush to the walls, 4 table-like boxes on the island inside the walk, 8 shelves, 8 lamps; everything keeps
    0.35 m clear of the camera path (1 m inset, 1.5 m high)."""
    state = [(int(seed) * 2654435761 + 12345) & 0xFFFFFFFF]

    def rnd():
        s = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
        s ^= s >> 15
        state[0] = s
        return (s >> 8) / 16777216.0

    rx, ry, rz = (float(v) for v in room)
    lo, hi = [], []

    def put(x0, y0, z0, x1, y1, z1):
        lo.append((x0, y0, z0))
        hi.append((x1, y1, z1))

    def on_wall(wall, a0, width, depth, z0, z1):
        if wall == 0:
            put(a0, 0, z0, a0 + width, depth, z1)
        elif wall == 1:
            put(rx - depth, a0, z0, rx, a0 + width, z1)
        elif wall == 2:
            put(a0, ry - depth, z0, a0 + width, ry, z1)
        else:
            put(0, a0, z0, depth, a0 + width, z1)

    for wall in range(4):
        length = ry if wall & 1 else rx
        for _ in range(7):
            depth = 0.25 + 0.35 * rnd()
            width = 0.4 + 1.2 * rnd()
            height = 0.4 + 1.6 * rnd() * rnd()
            a0 = (length - width) * rnd()
            on_wall(wall, a0, width, depth, 0, height)
    ix0, ix1, iy0, iy1 = 1.4, rx - 1.4, 1.4, ry - 1.4
    for _ in range(4):
        w = (0.2 + 0.6 * rnd()) * (ix1 - ix0)
        d = (0.3 + 0.7 * rnd()) * (iy1 - iy0)
        h = 0.4 + 0.8 * rnd()
        x0 = ix0 + (ix1 - ix0 - w) * rnd()
        y0 = iy0 + (iy1 - iy0 - d) * rnd()
        put(x0, y0, 0, x0 + w, y0 + d, h)
    for jj in range(8):
        j = jj & 3
        length = ry if j & 1 else rx
        depth = 0.05 + 0.25 * rnd()
        width = 0.5 + 1.0 * rnd()
        z0 = 1.2 + 0.7 * rnd()
        th = 0.05 + 0.25 * rnd()
        a0 = (length - width) * rnd()
        on_wall(j, a0, width, depth, z0, z0 + th)
    for _ in range(8):
        w = 0.3 + 0.5 * rnd()
        h = 0.15 + 0.3 * rnd()
        x0 = (rx - w) * rnd()
        y0 = (ry - w) * rnd()
        put(x0, y0, rz - h, x0 + w, y0 + w, rz)
    return np.asarray(lo, np.float32), np.asarray(hi, np.float32)


def _hash32(s):
    """lowbias32 on a uint64 array holding 32-bit values (csrc/synth.hip synth_hash)."""
    m = np.uint64(0xFFFFFFFF)
    s = s & m
    s ^= s >> np.uint64(16)
    s = (s * np.uint64(0x7feb352d)) & m
    s ^= s >> np.uint64(15)
    s = (s * np.uint64(0x846ca68b)) & m
    s ^= s >> np.uint64(16)
    return s


def render_room_depth(pose, width=640, height=480, room=ROOM, noise_frame=None, noise=1, boxes=None):
    """Analytic depth (u16 mm) of an axis-aligned box room [0,rx]x[0,ry]x[0,rz] seen from inside, optionally furnished with `boxes`
    (clutter_boxes(): scene 1 of csrc/synth.hip).

    noise_frame: if not None, adds 3 LSBs of noise for that frame index -- noise=1: the SURVEY 8d LCG step seeded with frame*W*H + pixel (a
    ramp: neighbouring seeds give neighbouring values; kept for the committed digests), noise=2: hashed per pixel and frame (what real low
    bits look like to a compressor) plus, in a furnished scene, sensor holes (grazing incidence below 0.12, 0.4 % speckle).
    """
    fx, fy, mx, my = intrinsics(width, height)
    u = (np.arange(width, dtype=np.float64) - mx) / fx
    v = (np.arange(height, dtype=np.float64) - my) / fy
    dc = np.stack(np.broadcast_arrays(u[None, :], v[:, None], np.ones((height, width))), -1)  # cam dirs, z = 1
    R = pose[:3, :3].astype(np.float64)
    o = pose[:3, 3].astype(np.float64)
    dw = dc @ R.T
    t = np.full((height, width), np.inf)
    axis = np.zeros((height, width), np.int64)   # normal of the surface hit
    for ax in range(3):
        d = dw[..., ax]
        with np.errstate(divide="ignore", invalid="ignore"):
            t_hi = np.where(d > 0, (room[ax] - o[ax]) / d, np.inf)
            t_lo = np.where(d < 0, (0.0 - o[ax]) / d, np.inf)
        ta = np.minimum(t_hi, t_lo)
        axis = np.where(ta < t, ax, axis)
        t = np.minimum(t, ta)
    if boxes is not None:
        for lo, hi in zip(np.asarray(boxes[0], np.float64), np.asarray(boxes[1], np.float64)):
            tn = np.zeros((height, width))
            tf = np.full((height, width), np.inf)
            an = np.zeros((height, width), np.int64)
            miss = np.zeros((height, width), bool)
            for ax in range(3):
                d = dw[..., ax]
                with np.errstate(divide="ignore", invalid="ignore"):
                    t0, t1 = (lo[ax] - o[ax]) / d, (hi[ax] - o[ax]) / d
                par = d == 0
                miss |= par & ((o[ax] < lo[ax]) | (o[ax] > hi[ax]))
                a0, a1 = np.where(par, -np.inf, np.minimum(t0, t1)), np.where(par, np.inf, np.maximum(t0, t1))
                an = np.where(a0 > tn, ax, an)
                tn = np.maximum(tn, a0)
                tf = np.minimum(tf, a1)
            hit = ~miss & (tn < tf) & (tn > 0) & (tn < t)
            t = np.where(hit, tn, t)
            axis = np.where(hit, an, axis)
    mm = np.rint(t * 1000.0)
    mm = np.where(np.isfinite(mm) & (mm < 65535), mm, 0).astype(np.int64)
    if noise_frame is not None:
        pix = np.arange(width * height, dtype=np.uint64).reshape(height, width)
        s = (np.uint64(noise_frame) * np.uint64(width * height) + pix) & np.uint64(0xFFFFFFFF)
        if noise == 1:
            s = (s * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
            mm = np.where(mm > 0, mm + ((s >> np.uint64(24)) & np.uint64(7)).astype(np.int64), 0)
        else:
            h = _hash32(s)
            mm = np.where(mm > 0, mm + (h >> np.uint64(29)).astype(np.int64), 0)
            if boxes is not None:
                cosi = np.abs(np.take_along_axis(dw, axis[..., None], -1)[..., 0]) / np.linalg.norm(dw, axis=-1)
                mm = np.where((cosi < 0.12) | ((h & np.uint64(0xFF)) == 0), 0, mm)
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


SCENE_DEFAULT = 1   # furnished room
NOISE_DEFAULT = 2   # hashed per pixel and frame


def render_scan_device(d_depth, stride, first, n, total, width, height, room=ROOM, origin=(0.0, 0.0, 0.0), noise=NOISE_DEFAULT, scene=SCENE_DEFAULT, seed=0):
    """Render frames into device memory (sf_synth_scene_device, scanfuse_internal.h); returns the n poses [n, 16] float32.
    noise: 0 none / 1 (or True) the round-1 LCG ramp / 2 hashed; scene: 0 empty box room / 1 furnished with clutter_boxes(room, seed)."""
    import ctypes as C
    from . import _abi
    L = _abi.lib()
    L.sf_synth_scene_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    poses = np.zeros((n, 16), np.float32)
    r = np.asarray(room, np.float64)
    o = np.asarray(origin, np.float64)
    _abi.check(L.sf_synth_scene_device(C.c_void_p(int(d_depth)), int(stride), int(first), int(n), int(total), int(width), int(height), int(noise),
                                       int(scene), int(seed) & 0xFFFFFFFF, r.ctypes.data, o.ctypes.data, poses.ctypes.data))
    return poses


def textured_pictures(cw, ch, count=8, amp=3.0):
    """Synthetic colour pictures with the entropy of real ones: a smooth scene under two octaves of sensor-like noise.  At quality 90, 4:2:0, a 1296x968
    picture compresses to ~190-205 KB -- ScanNet's own range is 50-250 KB (sensorData.h:600-616; SURVEY 8a row a3 probed 204 KB); the smooth
    pictures of round 4 came to 77-110 KB, and Huffman decoding costs per entropy-coded byte."""
    yy, xx = np.mgrid[0:ch, 0:cw]
    out = []
    for k in range(count):
        rng = np.random.default_rng(k)
        base = np.stack([(xx // 3 + 31 * k) % 256, (yy // 2 + 17 * k) % 256, (128 + 100 * np.sin(xx / 30.0) * np.cos(yy / 20.0 + k))], -1)
        coarse = rng.normal(0, amp, (ch // 2 + 1, cw // 2 + 1, 3)).repeat(2, 0).repeat(2, 1)[:ch, :cw]
        out.append(np.clip(base + coarse + rng.normal(0, amp / 2, (ch, cw, 3)), 0, 255).astype(np.uint8))
    return out
