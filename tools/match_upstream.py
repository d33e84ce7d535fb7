#!/usr/bin/env python3
"""Which upstream semantics does a given `_vh.ply` follow?  (INTEGRATION.md "Which upstream semantics".)

The fusion binaries the reference pipeline calls (DepthSensing.exe / FriedLiver.exe) are external; where their public sources are remembered to
differ from SURVEY App. C, `sf_params` has a switch.  A maintainer who holds the binaries fuses ONE scan with them and runs

    python tools/match_upstream.py <scan>.sens <scan>_vh.ply [--params zParametersScanNet.txt] [--frames N] [--device 0]

The tool fuses the same `.sens` on the GPU under every combination of the switches, extracts the mesh and scores it against the given PLY:
symmetric nearest-vertex distance (mean / 99th percentile, metres), the share of vertices without a counterpart within half a voxel, and the
mean colour difference at matched vertices.  The best-scoring line names the `s_scanfuse*` keys to put into the parameter file.
"""
import argparse
import itertools
import json
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannet_amd import fusion, sens  # noqa: E402
from scannet_amd.segmentator import Mesh  # noqa: E402

SWITCHES = ("frustum_mode", "colour_round", "colour_first", "weight_mode", "weight_wrap")
KEYS = {"frustum_mode": "s_scanfuseFrustumMode", "colour_round": "s_scanfuseColourRound", "colour_first": "s_scanfuseColourFirst",
        "weight_mode": "s_scanfuseWeightMode", "weight_wrap": "s_scanfuseWeightWrap"}


def score(xyz, rgb, ref_xyz, ref_rgb, voxel):
    """Symmetric nearest-vertex statistics of two vertex clouds (marching-cubes vertices sit on grid edges: equal semantics give equal sets)."""
    if len(xyz) == 0 or len(ref_xyz) == 0:
        return {"mean_m": float("inf"), "p99_m": float("inf"), "unmatched": 1.0, "colour": float("inf"), "vertices": int(len(xyz))}
    a, b = cKDTree(xyz), cKDTree(ref_xyz)
    d_ab, i_ab = b.query(xyz)
    d_ba, _ = a.query(ref_xyz)
    d = np.concatenate([d_ab, d_ba])
    near = d_ab < 0.5 * voxel
    col = float(np.abs(rgb[near].astype(np.int64) - ref_rgb[i_ab[near]].astype(np.int64)).mean()) if near.any() and ref_rgb is not None else 0.0
    return {"mean_m": float(d.mean()), "p99_m": float(np.percentile(d, 99)), "unmatched": float((d >= 0.5 * voxel).mean()), "colour": col,
            "vertices": int(len(xyz))}


def fuse(sd, params, switches, frames, device, weight_max_file):
    p = fusion.SfParams.from_buffer_copy(params)
    for k, v in switches.items():
        setattr(p, k, v)
    if switches.get("weight_wrap"):
        p.weight_max = weight_max_file      # the limit as the parameter file gives it (99999999 as shipped); sf_fuser_create keeps it for the wrap
    with fusion.Fuser(p, device=device) as f:
        f.run(sd, 0, frames)
        m = f.extract_mesh()
    xyz, rgba, _ = m.arrays()
    return xyz, rgba[:, :3]


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("sens")
    ap.add_argument("reference_ply")
    ap.add_argument("--params", default="", help="the parameter file the reference run used (zParametersScanNet.txt)")
    ap.add_argument("--frames", type=int, default=0, help="fuse only the first N frames (the reference PLY must come from the same frames)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--fix", action="append", default=[], metavar="SWITCH=0|1", help="do not vary this switch")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    sd = sens.SensorData(a.sens)
    p = fusion.default_params()
    weight_max_file = 255
    if a.params:
        p = fusion.load_params(a.params, base=p)
        weight_max_file = int(p.weight_max)
    p.depth_width, p.depth_height = sd.depth_width, sd.depth_height
    K = sd.intrinsic_depth
    p.fx, p.fy, p.mx, p.my = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    p.depth_shift = float(sd.depth_shift)
    ref = Mesh.read(a.reference_ply)
    ref_xyz, ref_rgba, _ = ref.arrays()
    fixed = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.fix}
    free = [k for k in SWITCHES if k not in fixed]
    rows = []
    for bits in itertools.product((0, 1), repeat=len(free)):
        sw = dict(fixed, **dict(zip(free, bits)))
        xyz, rgb = fuse(sd, p, sw, a.frames, a.device, weight_max_file)
        rows.append((sw, score(xyz, rgb, ref_xyz, ref_rgba[:, :3], p.voxel_size)))
    rows.sort(key=lambda r: (r[1]["unmatched"], r[1]["mean_m"], r[1]["colour"]))
    print("%-64s %10s %10s %10s %8s %9s" % ("switches", "unmatched", "mean [m]", "p99 [m]", "colour", "vertices"))
    for sw, sc in rows:
        name = " ".join("%s=%d" % (k, sw[k]) for k in SWITCHES)
        print("%-64s %10.5f %10.2e %10.2e %8.3f %9d" % (name, sc["unmatched"], sc["mean_m"], sc["p99_m"], sc["colour"], sc["vertices"]))
    best = rows[0][0]
    print("\nbest match -- parameter file lines:")
    for k in SWITCHES:
        print("%s = %d;" % (KEYS[k], best[k]))
    if a.json:
        json.dump([{"switches": sw, "score": sc} for sw, sc in rows], open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
