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

# name, (W, H), voxel, frame indices of the 1200-frame walk, colour?, deintegrate index (or None)
SCENARIOS = [
    ("room_8mm", (160, 120), 0.008, [0, 1, 2, 300, 301, 600], False, None),
    ("room_colour_deint", (96, 72), 0.02, [0, 40, 80, 120], True, 1),
    ("plane_4mm", (128, 96), 0.004, [], False, None),
]


def frames_of(name, size, idx, colour):
    W, H = size
    rng = np.random.default_rng(sum(map(ord, name)))
    out = []
    if not idx:
        out.append((synth.plane_frame(W, H, 1500), np.eye(4, dtype=np.float32), None))
    for i in idx:
        pose = synth.trajectory_pose(i, 1200)
        d = synth.render_room_depth(pose, W, H, noise_frame=i)
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if colour else None
        out.append((d, pose, rgb))
    return out


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run(volume_factory, name, size, voxel, idx, colour, deint):
    """volume_factory(W, H, voxel) -> object with integrate / deintegrate(depth, pose, rgb=) , export() -> (coords, voxels) sorted by
    (x, y, z), extract_mesh() -> (pos f32 [n,3], rgb u8 [n,3], tris int32 [m,3], keys u64 [n]) in canonical order."""
    W, H = size
    vol = volume_factory(W, H, voxel)
    fr = frames_of(name, size, idx, colour)
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

    def factory(W, H, voxel):
        p = orc.default_params(W, H, voxel)
        fx, fy, mx, my = synth.intrinsics(W, H)
        p.fx, p.fy, p.mx, p.my = fx, fy, mx, my
        vol = orc.Volume(p, threads=4)
        raw = vol.extract_mesh

        def extract():
            m = raw()
            return m["pos"], m["col"], m["idx"].astype(np.int32), m["keys"]
        vol.extract_mesh = extract
        return vol

    out = {name: run(factory, name, size, voxel, idx, colour, deint) for name, size, voxel, idx, colour, deint in SCENARIOS}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
