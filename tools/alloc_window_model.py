#!/usr/bin/env python3
"""How often does a ray leave k_alloc_ray's ray-space window?  (numpy model of the map in csrc/fuser.hip `anchor` / `win_bit`.)

For passes of 16 consecutive frames of the bench walk (furnished scene, hashed noise) the tool lays the window of every 16x16 pixel tile along the
mean of the tile's centre rays in the first and the last frame -- exactly the integers the kernel computes: dominant axis, k0, 12-bit fixed-point
slopes and offsets -- and tests, for every pixel of every frame, whether the two END blocks of its ray segment [d - t, d + t] fall inside the
16 x 16 x 256 window.  (The walk between two blocks inside a window that follows the pencil stays inside; the end blocks are where it leaves.)
Rays outside take the kernel's slow path (LDS hash set, then the global table): correct, slower.

  python tools/alloc_window_model.py [first frames of the passes ...]      default: 0 116 232 1000 1394 2789 4000
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannet_amd import synth  # noqa: E402

W, H, VOXEL, MAXD, TB, TS = 640, 480, 0.004, 4.0, 0.06, 0.02
BS = 8 * VOXEL
LAT, DEPTH = 16, 256


def centre_ray(pose, tx, ty, fx, fy, mx, my):
    kx, ky = (tx * 16 + 7.5 - mx) / fx, (ty * 16 + 7.5 - my) / fy
    R, t = pose[:3, :3].astype(np.float64), pose[:3, 3].astype(np.float64)
    return R @ np.array([kx, ky, 1.0]), t / BS


def window(pose_a, pose_b, tx, ty, intr):
    da, oa = centre_ray(pose_a, tx, ty, *intr)
    db, ob = centre_ray(pose_b, tx, ty, *intr)
    d, o = 0.5 * (da + db), 0.5 * (oa + ob)
    a = int(np.argmax(np.abs(d)))
    u, v = (a + 1) % 3, (a + 2) % 3
    sgn = -1 if d[a] < 0 else 1
    k0 = int(np.floor(o[a])) - sgn
    inv = 1.0 / d[a]
    a0 = (k0 + 0.5) - o[a]
    iu, iv = o[u] + d[u] * inv * a0 - LAT / 2, o[v] + d[v] * inv * a0 - LAT / 2
    return dict(a=a, u=u, v=v, sgn=sgn, k0=k0, su=int(np.rint(d[u] * inv * sgn * 4096)), sv=int(np.rint(d[v] * inv * sgn * 4096)),
                ou=int(np.floor(iu)), fu=int((iu - np.floor(iu)) * 4096), ov=int(np.floor(iv)), fv=int((iv - np.floor(iv)) * 4096))


def inside(w, blocks):
    k = (blocks[..., w["a"]] - w["k0"]) * w["sgn"]
    du = blocks[..., w["u"]] - w["ou"] - ((w["su"] * k + w["fu"]) >> 12)
    dv = blocks[..., w["v"]] - w["ov"] - ((w["sv"] * k + w["fv"]) >> 12)
    return (k >= 0) & (k < DEPTH) & (du >= 0) & (du < LAT) & (dv >= 0) & (dv < LAT)


def main():
    firsts = [int(a) for a in sys.argv[1:]] or [0, 116, 232, 1000, 1394, 2789, 4000]
    intr = synth.intrinsics(W, H)
    fx, fy, mx, my = intr
    boxes = synth.clutter_boxes()
    ys, xs = np.mgrid[0:H, 0:W]
    kx, ky = (xs - mx) / fx, (ys - my) / fy
    print("%-8s %12s %12s %14s" % ("pass at", "rays", "outside", "share"))
    for f0 in firsts:
        poses = [synth.trajectory_pose(f0 + j, 5578) for j in range(16)]
        depths = [synth.render_room_depth(poses[j], W, H, noise_frame=f0 + j, noise=2, boxes=boxes).astype(np.float64) / 1000.0 for j in range(16)]
        wins = {(tx, ty): window(poses[0], poses[15], tx, ty, intr) for ty in range(H // 16) for tx in range(W // 16)}
        rays = out = 0
        for j in range(16):
            d = depths[j]
            ok = (d > 0.1) & (d < MAXD)
            t = TS * d + TB
            R, tr = poses[j][:3, :3].astype(np.float64), poses[j][:3, 3].astype(np.float64)
            for z in (np.minimum(MAXD, d - t), np.minimum(MAXD, d + t)):
                pc = np.stack([kx * z, ky * z, z], -1)
                pw = pc @ R.T + tr
                blk = np.floor(np.floor(pw / VOXEL + 0.5) / 8).astype(np.int64)      # worldToBlock (round half up is near enough for a count)
                for (tx, ty), w in wins.items():
                    sl = (slice(ty * 16, ty * 16 + 16), slice(tx * 16, tx * 16 + 16))
                    m = ok[sl]
                    ins = inside(w, blk[sl])
                    rays += int(m.sum())
                    out += int((m & ~ins).sum())
        print("%-8d %12d %12d %13.4f%%" % (f0, rays // 2, out, 100.0 * out / max(rays, 1)))


if __name__ == "__main__":
    main()
