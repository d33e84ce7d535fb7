"""sf_mesh_clean_gpu (scannet_amd/csrc/clean_gpu.hip): the clean.mlx filters on the GPU must give the arrays and statistics of the host
filters -- and of the independent numpy/scipy restatement oracle/clean_oracle.py -- exactly: greedy non-transitive clustering (chains),
duplicate and flipped faces, components at the size threshold, unreferenced vertices, radius 0, empty inputs, a scan-sized mesh."""
import time

import numpy as np
import pytest

from scannet_amd import meshclean
from scannet_amd.segmentator import Mesh
from tests import meshes
from tests.test_meshclean import _grid, _soup

pytestmark = pytest.mark.gpu


def _same(a, b):
    (ax, ac, at), (bx, bc, bt) = a.arrays(), b.arrays()
    assert np.array_equal(ax.view(np.uint32), bx.view(np.uint32)) and np.array_equal(at, bt)
    assert (ac is None and bc is None) or np.array_equal(ac, bc)


@pytest.mark.parametrize("seed,min_cc", [(0, 100), (1, 10), (2, 2000), (3, 1)])
def test_soup_identical_to_host_and_to_the_restatement(seed, min_cc):
    from oracle import clean_oracle
    xyz, rgba, tris = _soup(np.random.default_rng(seed))
    m = Mesh.from_arrays(xyz, tris, rgba)
    host, hst = meshclean.clean(m, meshclean.CLEAN_MLX_MERGE_DISTANCE, min_cc)
    gpu, gst = meshclean.clean(m, meshclean.CLEAN_MLX_MERGE_DISTANCE, min_cc, gpu=0)
    _same(host, gpu)
    assert gst == hst and gst["vertices_merged"] >= 300 and gst["faces_duplicate"] >= 30
    ox, oc, ot = clean_oracle.clean(xyz, rgba, tris, meshclean.CLEAN_MLX_MERGE_DISTANCE, min_cc)
    gx, gc, gt = gpu.arrays()
    assert np.array_equal(gx.view(np.uint32), ox.view(np.uint32)) and np.array_equal(gc, oc) and np.array_equal(gt, ot)


def test_known_answers_gpu():
    # greedy, non-transitive clustering on a chain with 0.0008 spacing and radius 0.0010689: centres 0, 2, 4, 6 -- three rounds down the chain
    chain = np.array([[0.0008 * i, 0, 0] for i in range(7)], np.float32)
    apex = np.array([[0, 1, 0], [0, 1, 1]], np.float32)
    xyz = np.concatenate([chain, apex])
    tris = np.array([(i, 7, 8) for i in range(7)], np.uint32)
    out, st = meshclean.clean(Mesh.from_arrays(xyz, tris), 0.0010689, 0, gpu=0)
    gx, _, gt = out.arrays()
    assert np.array_equal(gx, xyz[[0, 2, 4, 6, 7, 8]])
    assert gt.tolist() == [[0, 4, 5], [1, 4, 5], [2, 4, 5], [3, 4, 5]]
    assert st["vertices_merged"] == 3 and st["faces_duplicate"] == 3 and st["faces_degenerate"] == 0
    # a long chain in index order: every vertex waits for the one before it (the worst case of the round-by-round resolution)
    n = 400
    chain = np.array([[0.0008 * i, 0, 0] for i in range(n)], np.float32)
    xyz = np.concatenate([chain, apex])
    tris = np.array([(i, n, n + 1) for i in range(n)], np.uint32)
    m = Mesh.from_arrays(xyz, tris)
    host, hst = meshclean.clean(m, 0.0010689, 0)
    gpu, gst = meshclean.clean(m, 0.0010689, 0, gpu=0)
    _same(host, gpu)
    assert gst == hst and gst["vertices_merged"] == n // 2
    # strict '<' at exactly the threshold distance
    r = np.float32(0.0010689)
    xyz = np.array([[0, 0, 0], [r, 0, 0], [0, 1, 0], [0, 1, 1]], np.float32)
    out, st = meshclean.clean(Mesh.from_arrays(xyz, np.array([(0, 2, 3), (1, 2, 3)], np.uint32)), float(r), 0, gpu=0)
    assert st["vertices_merged"] == 0 and st["faces_out"] == 2
    # 'fewer than': a component with exactly the minimum survives; radius 0 merges bit-identical positions only (-0.0 == +0.0)
    xyz, tris = _grid(3, 3)
    for min_cc, keep in ((8, 8), (9, 0)):
        out, st = meshclean.clean(Mesh.from_arrays(xyz, tris), 0.0, min_cc, gpu=0)
        assert st["faces_out"] == keep and st["vertices_out"] == (9 if keep else 0)
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [-0.0, 0, 0], [1, 0, 0], [0, 1, 1e-7]], np.float32)
    tris = np.array([(0, 1, 2), (3, 4, 5), (4, 3, 2)], np.uint32)
    m = Mesh.from_arrays(xyz, tris)
    host, hst = meshclean.clean(m, 0.0, 0)
    gpu, gst = meshclean.clean(m, 0.0, 0, gpu=0)
    _same(host, gpu)
    assert gst == hst and gst["vertices_merged"] == 2 and gst["faces_duplicate"] == 1
    # empty mesh, and vertices without faces
    out, st = meshclean.clean(Mesh.from_arrays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32)), gpu=0)
    assert out.counts() == (0, 0)
    m = Mesh.from_arrays(np.array([[0, 0, 0], [0.0005, 0, 0], [1, 1, 1]], np.float32), np.zeros((0, 3), np.uint32))
    host, hst = meshclean.clean(m, 0.0010689, 0)
    gpu, gst = meshclean.clean(m, 0.0010689, 0, gpu=0)
    assert gst == hst and gpu.counts() == host.counts() == (0, 0)


def test_scan_sized_mesh_identical_and_faster():
    """977 k faces of a bumpy height field plus seams (un-welded corners within the merge distance), duplicated faces and a thousand small
    islands: identical to the host filters, which take an order of magnitude longer."""
    v, t = meshes.bumpy_large(700)
    rng = np.random.default_rng(5)
    t = t.copy()
    extra = []
    for k in rng.choice(t.size, 20000, replace=False):      # seams
        f, c = divmod(int(k), 3)
        d = rng.normal(size=3)
        d *= rng.uniform(0, 0.0009) / np.linalg.norm(d)
        extra.append(v[t[f, c]] + d.astype(np.float32))
        t[f, c] = len(v) + len(extra) - 1
    v = np.concatenate([v, np.array(extra, np.float32)])
    islands_v, islands_t = [], []
    for i in range(1000):                                    # 8-face islands far from the sheet
        gv, gt = _grid(3, 3, origin=(10.0 + 0.1 * i, 0, 0))
        islands_t.append(gt + len(v) + 9 * i)
        islands_v.append(gv)
    v = np.concatenate([v] + islands_v)
    t = np.concatenate([t, t[1000:3000], t[5000:6000][:, ::-1]] + islands_t).astype(np.uint32)
    rgba = rng.integers(0, 256, (len(v), 4), dtype=np.uint8)
    m = Mesh.from_arrays(v, t, rgba)
    t0 = time.perf_counter()
    host, hst = meshclean.clean(m, meshclean.CLEAN_MLX_MERGE_DISTANCE, 7500)
    t1 = time.perf_counter()
    meshclean.clean(m, meshclean.CLEAN_MLX_MERGE_DISTANCE, 7500, gpu=0)        # first call pays the module load
    t2 = time.perf_counter()
    gpu, gst = meshclean.clean(m, meshclean.CLEAN_MLX_MERGE_DISTANCE, 7500, gpu=0)
    t3 = time.perf_counter()
    _same(host, gpu)
    assert gst == hst
    assert gst["components_removed"] >= 1000 and gst["faces_duplicate"] == 3000 and gst["vertices_merged"] > 15000, gst
    print("clean %d faces: host %.3f s, gpu %.3f s" % (len(t), t1 - t0, t3 - t2))
    assert t3 - t2 < t1 - t0


def test_many_random_soups_gpu():
    """Differential run over 60 random triangle soups: clustered vertices (dense enough that chains of the greedy clustering are several
    vertices long), random duplicate and degenerate faces, components around the size threshold, merge distance 0 and > 0."""
    rng = np.random.default_rng(2024)
    for it in range(60):
        nv = int(rng.integers(50, 3000))
        centres = rng.uniform(0, 0.2, (max(nv // 4, 2), 3))
        v = (centres[rng.integers(0, len(centres), nv)] + rng.normal(0, 0.0007, (nv, 3)) * (rng.random((nv, 1)) < 0.7)).astype(np.float32)
        v[rng.random(nv) < 0.05] *= np.float32(-0.0)          # a few exact (signed) zeros
        nf = int(rng.integers(10, 6000))
        t = rng.integers(0, nv, (nf, 3)).astype(np.uint32)
        k = max(1, nf // 10)
        t[rng.integers(0, nf, k)] = t[rng.integers(0, nf, k)][:, rng.permutation(3)]     # duplicates with permuted corners
        t[rng.integers(0, nf, max(1, nf // 50)), 1] = t[rng.integers(0, nf, max(1, nf // 50)), 0]   # degenerate faces
        radius = float(rng.choice([0.0, 0.0005, 0.0010689, 0.003]))
        min_cc = int(rng.choice([0, 1, 2, 5, 50]))
        m = Mesh.from_arrays(v, t, rng.integers(0, 256, (nv, 4), dtype=np.uint8))
        host, hst = meshclean.clean(m, radius, min_cc)
        gpu, gst = meshclean.clean(m, radius, min_cc, gpu=0)
        _same(host, gpu)
        assert gst == hst, (it, radius, min_cc, gst, hst)
