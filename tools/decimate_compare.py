#!/usr/bin/env python3
"""How far the two decimation variants land from each other (Weak 2 / Next 7b).

The `decimate` stage (Server/scan_processor.py:149-153: simplify.mlx twice, each followed by cleanLoRes) exists twice in this library:
  * sequential   sf_mesh_simplify      (csrc/simplify.cpp): one priority queue, the greedy order MeshLab's filter uses (simplify.mlx:3-16);
  * GPU rounds   sf_mesh_simplify_gpu  (csrc/simplify_gpu.hip): rounds of independent collapses -- a different algorithm with the same guarantees.
Neither is pinned to MeshLab (absent here); this tool measures the distance BETWEEN them on the same input, and from each to the input:
  faces / vertices out, two-sided surface distance (area-weighted samples of one surface against dense samples of the other: max = sampled Hausdorff,
  mean, RMS, 99.9th percentile), and what Segmentator makes of each result: segment count and the adjusted Rand index of the two labelings carried to
  common sample points of the INPUT surface (each sample takes the label of the nearest vertex of the decimated mesh).

  python tools/decimate_compare.py [--mesh bumpy|room] [--n 700] [--frames 400] [--out decimate_compare.json]
needs a GPU (the rounds variant has no CPU path)."""
import argparse
import json
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def surface_samples(v, t, n, rng):
    """n points on the surface, area-weighted, with the face each came from."""
    a, b, c = v[t[:, 0]].astype(np.float64), v[t[:, 1]].astype(np.float64), v[t[:, 2]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    p = area / area.sum()
    f = rng.choice(len(t), size=n, p=p)
    r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
    w0, w1, w2 = 1 - r1, r1 * (1 - r2), r1 * r2
    return a[f] * w0[:, None] + b[f] * w1[:, None] + c[f] * w2[:, None], f, float(area.sum())


def one_sided(src_pts, dst_v, dst_t, rng, dense):
    """Distances from src_pts to the surface (dst_v, dst_t), the surface approximated by `dense` area-weighted samples plus its vertices: an upper
    bound on the true point-to-surface distance that is off by at most the dense sample spacing."""
    dpts, _, area = surface_samples(dst_v, dst_t, dense, rng)
    tree = cKDTree(np.concatenate([dpts, dst_v.astype(np.float64)]))
    d, _ = tree.query(src_pts, workers=-1)
    return d, float(np.sqrt(area / dense))


def dist_stats(d):
    return {"max": float(d.max()), "mean": float(d.mean()), "rms": float(np.sqrt((d * d).mean())), "p999": float(np.quantile(d, 0.999))}


def two_sided(va, ta, vb, tb, rng, n=300000, dense=3000000):
    pa, _, _ = surface_samples(va, ta, n, rng)
    pb, _, _ = surface_samples(vb, tb, n, rng)
    dab, sp_b = one_sided(pa, vb, tb, rng, dense)
    dba, sp_a = one_sided(pb, va, ta, rng, dense)
    return {"a_to_b": dist_stats(dab), "b_to_a": dist_stats(dba), "hausdorff_sampled": float(max(dab.max(), dba.max())),
            "dense_sample_spacing": float(max(sp_a, sp_b))}


def decimate_twice(mesh, gpu, meshclean):
    """simplify.mlx + cleanLoRes, twice (scannet_amd/shard.py finish_scan)."""
    cur, secs, stats = mesh, 0.0, []
    for _ in range(2):
        t0 = time.perf_counter()
        simp, st = meshclean.simplify(cur, gpu=gpu)
        cur, _ = meshclean.clean(simp, meshclean.CLEAN_MLX_MERGE_DISTANCE, meshclean.CLEAN_LORES_MIN_COMPONENT)
        secs += time.perf_counter() - t0
        stats.append({k: (float(v) if isinstance(v, float) else int(v)) for k, v in st.items()})
    return cur, secs, stats


def compare(v, t, rng, label, n_samples=300000, dense=3000000):
    from scannet_amd import meshclean, segmentator
    from scannet_amd.segmentator import Mesh
    m = Mesh.from_arrays(v, t)
    seq, t_seq, st_seq = decimate_twice(m, None, meshclean)
    gpu, t_gpu, st_gpu = decimate_twice(m, 0, meshclean)
    sv, _, stt = seq.arrays()
    gv, _, gtt = gpu.arrays()
    out = {"input": {"mesh": label, "vertices": int(len(v)), "faces": int(len(t))},
           "sequential": {"vertices": int(len(sv)), "faces": int(len(stt)), "seconds": round(t_seq, 2), "passes": st_seq},
           "gpu_rounds": {"vertices": int(len(gv)), "faces": int(len(gtt)), "seconds": round(t_gpu, 2), "passes": st_gpu}}
    out["faces_ratio_gpu_over_sequential"] = round(len(gtt) / max(len(stt), 1), 4)
    out["sequential_vs_gpu"] = two_sided(sv, stt, gv, gtt, rng, n_samples, dense)
    ref_used = v[np.unique(t)]
    for name, (xv, xt) in (("sequential", (sv, stt)), ("gpu_rounds", (gv, gtt))):
        out[name]["vs_input"] = two_sided(xv, xt, v, t, rng, n_samples, dense)
        seg = segmentator.segment_arrays(xv, xt, 0.01, 20)
        out[name]["segments"] = int(len(np.unique(seg)))
        out[name]["_seg"] = seg
    # both labelings carried to common points of the input surface
    pts, _, _ = surface_samples(v, t, n_samples, rng)
    la = out["sequential"].pop("_seg")[cKDTree(sv.astype(np.float64)).query(pts, workers=-1)[1]]
    lb = out["gpu_rounds"].pop("_seg")[cKDTree(gv.astype(np.float64)).query(pts, workers=-1)[1]]
    from sklearn.metrics import adjusted_rand_score, normalized_mutual_info_score
    out["segmentation_agreement"] = {"adjusted_rand_index": round(float(adjusted_rand_score(la, lb)), 4),
                                     "normalized_mutual_information": round(float(normalized_mutual_info_score(la, lb)), 4),
                                     "segments_ratio_gpu_over_sequential": round(out["gpu_rounds"]["segments"] / max(out["sequential"]["segments"], 1), 4),
                                     "samples": int(n_samples),
                                     "what": "Segmentator (kThresh 0.01, segMinVerts 20) on each decimated mesh; each of %d area-weighted points of the INPUT "
                                             "surface takes the label of the nearest vertex of either result" % n_samples}
    del ref_used
    extent = float(np.linalg.norm(v[np.unique(t)].max(0) - v[np.unique(t)].min(0)))
    out["input"]["bbox_diagonal"] = extent
    for x in (m, seq, gpu):
        x.close()
    return out


def room_mesh(frames):
    """The furnished synthetic room fused on the GPU (4 mm voxels), marching cubes, clean.mlx: what the decimate stage really receives."""
    import torch
    from scannet_amd import fusion, meshclean, synth
    W, H = 640, 480
    dev = torch.empty((frames, H, W), dtype=torch.int16, device="cuda")
    poses = synth.render_scan_device(dev.data_ptr(), W * H * 2, 0, frames, 5578, W, H, noise=2, scene=1, seed=0)
    p = fusion.default_params()
    with fusion.Fuser(p, device=0) as f:
        f.integrate_batch_device(dev.data_ptr(), W * H * 2, poses)
        mesh = f.extract_mesh()
    cleaned, _ = meshclean.clean(mesh, meshclean.CLEAN_MLX_MERGE_DISTANCE, meshclean.CLEAN_MLX_MIN_COMPONENT)
    v, _, t = cleaned.arrays()
    mesh.close()
    return np.ascontiguousarray(v), np.ascontiguousarray(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", choices=["bumpy", "room", "both"], default="both")
    ap.add_argument("--n", type=int, default=700)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--samples", type=int, default=300000)
    ap.add_argument("--dense", type=int, default=3000000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    res = {}
    if a.mesh in ("bumpy", "both"):
        from tests import meshes
        v, t = meshes.bumpy_large(a.n)
        res["bumpy"] = compare(np.asarray(v, np.float32), np.asarray(t, np.uint32), rng, "tests/meshes.py bumpy_large(%d)" % a.n, a.samples, a.dense)
    if a.mesh in ("room", "both"):
        v, t = room_mesh(a.frames)
        res["room"] = compare(v, t, rng, "furnished synthetic room, %d frames fused at 4 mm on the GPU, marching cubes, clean.mlx" % a.frames, a.samples, a.dense)
    txt = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
