#!/usr/bin/env python3
"""Upstream-conformance packet (INTEGRATION.md "Conformance packet"): everything a maintainer who holds the Windows fusion binary
needs to return the FIRST real golden vector for the external TSDF path -- and what this repository produces on the same input under every
combination of the five upstream-conformance switches.

    python tools/conformance_packet.py [--out conformance/] [--frames walk|full]      # CPU only (numpy renderer + oracle/): ~2 minutes
    python tools/conformance_packet.py --verify-gpu                                    # the HIP path against the committed digests (needs an MI355X)

The input is small and deterministic -- no RNG, generated from closed forms by scannet_amd/synth.py:
  * 40 frames of the furnished-room walk (every 24th frame of a 960-frame loop) at 320x240, depth in mm with 3 hashed noise bits + sensor holes,
  * then ONE view held for 260 more frames (the 8-bit weight passes 255: the only place the weight_wrap switch shows),
  * a raw RGB frame per depth frame (a moving gradient over a texture, a black band: colour_first; odd channel sums: colour_round),
  * .sens v4: zlib depth (one fixed-Huffman block, as the reference's stb writer emits), TYPE_RAW colour, poses, depthShift 1000.
`conformance.sens` is written next to the digests (22 MB; not committed -- the generator is); `digests.json` holds, per switch combination, the
sha256 of the fused volume (block coordinates + 8-byte voxels) and of the canonical mesh (vertex positions, colours, triangles), the block /
vertex / triangle counts, for the 40-frame walk alone and for all 300 frames.  The digests are the ORACLE's (oracle/tsdf_oracle.c, CPU); the GPU
tests reproduce them through the C ABI (tests/test_gpu_tsdf.py::test_conformance_packet_digests).
"""
import argparse
import hashlib
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scannet_amd import synth  # noqa: E402

W, H = 320, 240
VOXEL = 0.01
WALK, DWELL, LOOP = 40, 260, 960
SWITCHES = ("frustum_mode", "colour_round", "colour_first", "weight_mode", "weight_wrap")
WEIGHT_SAMPLE = 3          # > 1 so that weight_mode (depth-dependent observation weight) is visible; the shipped file has 1, where both modes agree
WEIGHT_MAX_FILE = 99999999  # zParametersScanNet.txt:53 as shipped


def colour_frame(i):
    """Closed form, uint8 [H, W, 3]: moving gradients over a fixed texture; the top eighth black; a saturated patch."""
    yy, xx = np.mgrid[0:H, 0:W]
    tex = ((xx * 7919 + yy * 104729) >> 3) & 31
    out = np.stack([(xx * 255 // W + 5 * i + tex) % 256, (yy * 255 // H + 3 * i + tex) % 256, (xx + yy + 7 * i + tex) % 256], -1).astype(np.uint8)
    out[: H // 8] = 0
    out[H // 2: H // 2 + H // 16, : W // 3] = 255
    return out


def frames(which="full"):
    """-> list of (depth u16 [H, W], pose 4x4 f32, rgb u8 [H, W, 3]); `walk`: the first 40, `full`: all 300."""
    boxes = synth.clutter_boxes()
    out = []
    for k in range(WALK):
        i = k * (LOOP // WALK)
        pose = synth.trajectory_pose(i, LOOP)
        out.append((synth.render_room_depth(pose, W, H, noise_frame=i, noise=2, boxes=boxes), pose, colour_frame(k)))
    if which == "full":
        d, pose, _ = out[-1]
        for k in range(DWELL):
            out.append((d, pose, colour_frame(WALK + k)))   # the same depth image again: what the dwell adds is weight, not geometry
    return out


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def oracle_params(orc, sw):
    p = orc.default_params(W, H, VOXEL)
    fx, fy, mx, my = synth.intrinsics(W, H)
    p.fx, p.fy, p.mx, p.my = fx, fy, mx, my
    p.weight_sample = WEIGHT_SAMPLE
    for k, v in sw.items():
        setattr(p, k, v)
    p.weight_max = WEIGHT_MAX_FILE if sw.get("weight_wrap") else 255
    return p


def gpu_params(fusion, sw):
    fx, fy, mx, my = synth.intrinsics(W, H)
    p = fusion.default_params(depth_width=W, depth_height=H, voxel_size=VOXEL, fx=fx, fy=fy, mx=mx, my=my, num_sdf_blocks=1 << 17, weight_sample=WEIGHT_SAMPLE)
    for k, v in sw.items():
        setattr(p, k, v)
    p.weight_max = WEIGHT_MAX_FILE if sw.get("weight_wrap") else 255
    return p


def result_of(coords, vox, mesh):
    pos, col, idx = mesh
    return {"blocks": int(len(coords)), "volume_sha256": digest(coords, vox), "vertices": int(len(pos)), "triangles": int(len(idx)),
            "mesh_sha256": digest(pos, col, idx)}


def run_oracle(orc, fr, sw, threads):
    vol = orc.Volume(oracle_params(orc, sw), threads=threads)
    out = {}
    for k, (d, pose, rgb) in enumerate(fr):
        vol.integrate(d, pose, rgb=rgb)
        if k + 1 == WALK or k + 1 == len(fr):
            c, v = vol.export()
            m = vol.extract_mesh()
            out["walk" if k + 1 == WALK else "full"] = result_of(c, v, (m["pos"], m["col"], m["idx"]))
    vol.close()
    return out


def run_gpu(fusion, fr, sw, device=0):
    out = {}
    with fusion.Fuser(gpu_params(fusion, sw), device=device) as f:
        for k, (d, pose, rgb) in enumerate(fr):
            f.integrate(d, pose, rgb=rgb)
            if k + 1 == WALK or k + 1 == len(fr):
                c, v = f.export_blocks()
                xyz, rgba, tris = f.extract_mesh().arrays()
                out["walk" if k + 1 == WALK else "full"] = result_of(c, v, (xyz, rgba[:, :3], tris.astype(np.int32)))
    return out


def name_of(sw):
    return "".join(str(sw[k]) for k in SWITCHES)


def combos():
    for bits in itertools.product((0, 1), repeat=len(SWITCHES)):
        yield dict(zip(SWITCHES, bits))


PARAM_LINES = """// conformance/zParametersConformance.txt -- the lines to CHANGE in the parameter file of the upstream binary for the conformance scan
// (everything else as in Server/tools/recons/zParametersScanNet.txt); bin/depthsensing of this repository reads the same keys.
s_SDFVoxelSize = 0.010f;
s_SDFMarchingCubeThreshFactor = 10.0f;
s_SDFTruncation = 0.06f;
s_SDFTruncationScale = 0.02f;
s_SDFMaxIntegrationDistance = 4.0f;
s_SDFIntegrationWeightSample = %d;
s_SDFIntegrationWeightMax = %d;
s_sensorDepthMin = 0.1f;
s_sensorDepthMax = 6.0f;
s_integrationWidth = %d;
s_integrationHeight = %d;
s_hashNumBuckets = 200000;
s_hashNumSDFBlocks = 131072;
s_garbageCollectionEnabled = false;
""" % (WEIGHT_SAMPLE, WEIGHT_MAX_FILE, W, H)


def write_sens(path, fr):
    from scannet_amd import sens
    K = synth.intrinsic_matrix(W, H)
    sd = sens.SensorData.create(W, H, W, H, K, K, color_compression=0, depth_compression=1, sensor_name="StructureSensor")
    for k, (d, pose, rgb) in enumerate(fr):
        sd.add_frame(d, pose, color=rgb, timestamp_color=33333 * k, timestamp_depth=33333 * k)
    sd.save(path)
    sd.close()
    return hashlib.sha256(open(path, "rb").read()).hexdigest(), os.path.getsize(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "conformance"))
    ap.add_argument("--frames", choices=["walk", "full"], default="full")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 4)
    ap.add_argument("--verify-gpu", action="store_true", help="fuse the packet's frames on the GPU under all 32 combinations and compare with digests.json")
    ap.add_argument("--no-sens", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    fr = frames(a.frames)
    if a.verify_gpu:
        from scannet_amd import fusion
        want = json.load(open(os.path.join(a.out, "digests.json")))["combinations"]
        bad = 0
        for sw in combos():
            got = run_gpu(fusion, fr, sw)
            for part, g in got.items():
                ok = g == want[name_of(sw)][part]
                bad += not ok
                print(name_of(sw), part, "ok" if ok else "DIFFERS", g["blocks"], g["vertices"])
        raise SystemExit(1 if bad else 0)
    from oracle import oracle as orc
    out = {"what": __doc__.split("\n\n")[0], "switch_order": list(SWITCHES), "size": [W, H], "voxel": VOXEL, "frames": {"walk": WALK, "full": len(fr)},
           "weight_sample": WEIGHT_SAMPLE, "weight_max_in_file": WEIGHT_MAX_FILE,
           "presets": {"survey_app_c (default)": "00000", "voxelhashing (--upstream, s_scanfuseUpstream = 1)": "11111", "bundlefusion (s_scanfuseUpstream = 2)": "11101"},
           "combinations": {}}
    if not a.no_sens:
        sha, size = write_sens(os.path.join(a.out, "conformance.sens"), fr)
        out["sens"] = {"file": "conformance.sens (generated, not committed)", "sha256": sha, "bytes": size,
                       "note": "as generated in this repository's container (numpy %s); a platform whose libm rounds sin / cos differently may move single pixels" % np.__version__}
    for sw in combos():
        out["combinations"][name_of(sw)] = run_oracle(orc, fr, sw, a.threads)
        r = out["combinations"][name_of(sw)]
        print(name_of(sw), {k: (v["blocks"], v["vertices"], v["mesh_sha256"][:12]) for k, v in r.items()})
    json.dump(out, open(os.path.join(a.out, "digests.json"), "w"), indent=1)
    open(os.path.join(a.out, "zParametersConformance.txt"), "w").write(PARAM_LINES)
    distinct = len({v.get("full", v["walk"])["mesh_sha256"] for v in out["combinations"].values()})
    print("%d combinations, %d distinct meshes" % (len(out["combinations"]), distinct))


if __name__ == "__main__":
    main()
