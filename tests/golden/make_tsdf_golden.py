#!/usr/bin/env python3
"""Regenerates tests/golden/tsdf_golden.json: SHA-256 digests of what the CPU oracle (oracle/tsdf_oracle.c +
mc_oracle.c) produces on small seeded scenarios.  The reference holds no TSDF code, so these are NOT reference vectors
("parity unpinned"): they freeze the specification's executable form, so that neither the oracle nor the HIP path can
drift silently -- tests/test_oracle_tsdf.py checks the oracle against them on CPU, tests/test_gpu_tsdf.py the GPU path.

    python tests/golden/make_tsdf_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scannet_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tsdf_golden.json")

# name, (W, H), voxel, frame indices of the 1200-frame walk, colour?, deintegrate index (or None), options:
#   furnished: the room with the 48 boxes, hashed noise and sensor holes (scene 1 / noise 2 of bench.py) instead of the empty room with the LCG ramp
#   repeat:    every frame index is fused that many times with fresh noise (weights run into the clamp / the wrap)
#   switches:  sf_params / or_params conformance switches and weight limits
SCENARIOS = [
    ("room_8mm", (160, 120), 0.008, [0, 1, 2, 300, 301, 600], False, None, {}),
    ("room_colour_deint", (96, 72), 0.02, [0, 40, 80, 120], True, 1, {}),
    ("plane_4mm", (128, 96), 0.004, [], False, None, {}),
    # round 3: the furnished scene under the upstream-style switches, and 300 observations of two views into the 8-bit weight wrap
    ("furnished_switched", (128, 96), 0.016, list(range(0, 1200, 60)), True, 3,
     {"furnished": True, "switches": {"frustum_mode": 1, "colour_round": 1, "colour_first": 1, "weight_mode": 1, "weight_sample": 6}}),
    ("weight_wrap_300", (96, 72), 0.02, [50, 90], False, None,
     {"furnished": True, "repeat": 150, "switches": {"weight_wrap": 1, "weight_max": 99999999}}),
    ("clamp_300", (96, 72), 0.02, [50, 90], True, 0, {"furnished": True, "repeat": 150}),
]


def frames_of(name, size, idx, colour, opts=None):
    opts = opts or {}
    W, H = size
    rng = np.random.default_rng(sum(map(ord, name)))
    boxes = synth.clutter_boxes() if opts.get("furnished") else None
    out = []
    if not idx:
        out.append((synth.plane_frame(W, H, 1500), np.eye(4, dtype=np.float32), None))
    k = 0
    for _ in range(opts.get("repeat", 1)):
        for i in idx:
            pose = synth.trajectory_pose(i, 1200)
            if boxes is not None:
                d = synth.render_room_depth(pose, W, H, noise_frame=1000 * k + i, noise=2, boxes=boxes)
            else:
                d = synth.render_room_depth(pose, W, H, noise_frame=i)
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if colour else None
            if rgb is not None and boxes is not None:
                rgb[:, : W // 4] = 0   # a black band (colour_first)
            out.append((d, pose, rgb))
            k += 1
    return out


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run(volume_factory, name, size, voxel, idx, colour, deint, opts=None):
    """volume_factory(W, H, voxel, **switches) -> object with integrate / deintegrate(depth, pose, rgb=) , export() -> (coords, voxels) sorted by
    (x, y, z), extract_mesh() -> (pos f32 [n,3], rgb u8 [n,3], tris int32 [m,3], keys u64 [n]) in canonical order."""
    opts = opts or {}
    W, H = size
    vol = volume_factory(W, H, voxel, **opts.get("switches", {}))
    fr = frames_of(name, size, idx, colour, opts)
    for d, pose, rgb in fr:
        vol.integrate(d, pose, rgb=rgb)
    if deint is not None:
        d, pose, rgb = fr[deint]
        vol.deintegrate(d, pose, rgb=rgb)
    coords, vox = vol.export()
    pos, col, tris, keys = vol.extract_mesh()
    return {"blocks": int(len(coords)), "voxels_sha256": digest(coords, vox.view(np.uint8)),
            "mesh_vertices": int(len(pos)), "mesh_faces": int(len(tris)), "mesh_sha256": digest(keys, pos, col, tris)}


def main():
    from oracle import oracle as orc

    def factory(W, H, voxel, **switches):
        p = orc.default_params(W, H, voxel)
        fx, fy, mx, my = synth.intrinsics(W, H)
        p.fx, p.fy, p.mx, p.my = fx, fy, mx, my
        for k, v in switches.items():
            setattr(p, k, v)
        vol = orc.Volume(p, threads=4)
        raw = vol.extract_mesh

        def extract():
            m = raw()
            return m["pos"], m["col"], m["idx"].astype(np.int32), m["keys"]
        vol.extract_mesh = extract
        return vol

    out = {name: run(factory, name, size, voxel, idx, colour, deint, opts) for name, size, voxel, idx, colour, deint, opts in SCENARIOS}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
