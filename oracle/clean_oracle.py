"""oracle/clean_oracle.py -- numpy/scipy restatement of the four MeshLab filters of Server/tools/meshclean/clean.mlx:3-10.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: neither MeshLab nor VCG is in the reference tree and no version is pinned
(Server/config.py:19 names an installed `VCG\\MeshLab\\meshlabserver.exe`); this file restates the published VCG
algorithms behind the filters (vcg/complex/algorithms/clean.h: ClusterVertex + RemoveDuplicateVertex,
RemoveDuplicateFace, RemoveSmallConnectedComponentsSize, RemoveUnreferencedVertex) independently of
scannet_amd/csrc/clean.cpp: a kd-tree instead of a hash grid, scipy's connected_components instead of a union-find.
Filter parameters are pinned by clean.mlx:4 (Threshold 0.0010689, absolute) and clean.mlx:8 (MinComponentSize 7500).
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree


def merge_close_vertices(xyz, radius):
    """ClusterVertex: visit vertices in index order; an unvisited vertex becomes a centre, every still unvisited vertex at
    float32 distance < radius moves onto it (greedy, not transitive).  Returns target[i] = index i is merged into."""
    xyz = np.asarray(xyz, np.float32)
    n = len(xyz)
    p = xyz.copy()
    visited = np.zeros(n, bool)
    if radius > 0 and n:
        tree = cKDTree(xyz.astype(np.float64))
        r32 = np.float32(radius)
        for i in range(n):
            if visited[i]:
                continue
            visited[i] = True
            c = p[i]
            for j in tree.query_ball_point(xyz[i].astype(np.float64), float(radius) * 1.0001 + 1e-12):
                if visited[j]:
                    continue
                e = c - p[j]                                           # float32, like vcg::Distance
                dist = np.sqrt(np.float32(np.float32(e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]))
                if dist < r32:
                    visited[j] = True
                    p[j] = c
    # RemoveDuplicateVertex: identical positions collapse into the lowest index
    target = np.arange(n)
    first = {}
    for i in range(n):
        key = (float(p[i, 0]), float(p[i, 1]), float(p[i, 2]))          # -0.0 == 0.0 hash alike
        target[i] = first.setdefault(key, i)
    return target


def clean(xyz, rgba, tris, merge_distance=0.0010689, min_component_faces=7500):
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    tris = np.asarray(tris, np.int64).reshape(-1, 3)
    target = merge_close_vertices(xyz, merge_distance)
    t = target[tris]
    t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]            # degenerate faces
    # duplicate faces: same vertex set, lowest face index survives
    s = np.sort(t, axis=1)
    _, first = np.unique(s, axis=0, return_index=True)
    t = t[np.sort(first)]
    # connected components over shared edges
    nf = len(t)
    if nf:
        e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
        e.sort(axis=1)
        f = np.tile(np.arange(nf), 3)
        order = np.lexsort((e[:, 1], e[:, 0]))
        e, f = e[order], f[order]
        same = (e[1:] == e[:-1]).all(axis=1)
        a, b = f[1:][same], f[:-1][same]
        g = coo_matrix((np.ones(len(a), np.int8), (a, b)), shape=(nf, nf))
        _, lab = connected_components(g, directed=False)
        size = np.bincount(lab)
        t = t[size[lab] >= min_component_faces]
    used = np.zeros(len(xyz), bool)
    used[t.ravel()] = True
    remap = np.cumsum(used) - 1
    out_rgba = None if rgba is None else np.asarray(rgba, np.uint8).reshape(-1, 4)[used]
    return xyz[used], out_rgba, remap[t].astype(np.uint32)
