"""oracle/simplify_rounds_oracle.c -- the sequential restatement of the GPU decimation's rule (rounds of independent collapses) -- on its own, without a
GPU: it builds, meets the face budget, keeps flat flat and closed closed, is deterministic, and its initial placement agrees with the known answers of
tests/test_simplify_known_answers.py's kind (a coplanar patch collapses at the floor priority and stays in its plane).  The identity with the GPU
implementation is tests/test_simplify_gpu.py::test_gpu_collapse_is_the_sequential_restatement_of_its_rule (needs the MI355X)."""
import numpy as np

from tests import meshes
from tests.test_simplify import _edge_counts, _icosphere, _plane


def test_budget_planarity_closedness(oracle):
    v, t = _plane(60)
    xyz, _, tris, st = oracle.simplify_rounds(v, t)
    target = int(len(t) * 0.2)
    assert target - 2 <= len(tris) <= target and st["rounds"] > 3 and st["collapses"] > 0
    assert np.abs(xyz[:, 2]).max() == 0.0 and xyz[:, :2].min() >= -1e-6 and xyz[:, :2].max() <= 1 + 1e-6
    a, b, c = xyz[tris[:, 0]], xyz[tris[:, 1]], xyz[tris[:, 2]]
    n = np.cross(b - a, c - a)
    assert (n[:, 2] > 0).all() and abs(0.5 * np.linalg.norm(n, axis=1).sum() - 1.0) < 1e-3
    v, t = _icosphere(4)
    xyz, _, tris, _ = oracle.simplify_rounds(v, t)
    assert len(tris) == int(len(t) * 0.2) and (_edge_counts(tris) == 2).all() and len(xyz) - len(tris) // 2 == 2
    assert abs(np.linalg.norm(xyz, axis=1) - 1).max() < 5e-3


def test_deterministic_and_colours_follow_the_surviving_vertex(oracle):
    v, t = meshes.bumpy(80)[:2]
    rgba = np.stack([np.arange(len(v)) % 256, (np.arange(len(v)) // 256) % 256, np.zeros(len(v)), np.full(len(v), 255)], -1).astype(np.uint8)
    x1, c1, t1, s1 = oracle.simplify_rounds(v, t, rgba)
    x2, c2, t2, s2 = oracle.simplify_rounds(v, t, rgba)
    assert np.array_equal(x1.view(np.uint32), x2.view(np.uint32)) and np.array_equal(t1, t2) and np.array_equal(c1, c2) and s1 == s2
    # a surviving vertex keeps ITS colour (v1's, as VCG does): the colour encodes the input index, which must be strictly increasing after compaction
    idx = c1[:, 0].astype(np.int64) + 256 * c1[:, 1].astype(np.int64)
    assert (np.diff(idx) > 0).all()


def test_parameters_reach_the_rule(oracle):
    v, t = _icosphere(3)
    for kw, want in (({"target_perc": 0.5}, len(t) // 2), ({"target_perc": 0.0, "target_faces": 300}, 300)):
        _, _, tris, _ = oracle.simplify_rounds(v, t, **kw)
        assert want - 2 <= len(tris) <= want
    xa, _, _, _ = oracle.simplify_rounds(v, t, optimal_placement=False)
    assert np.isin(xa.view(np.uint32).view(np.dtype((np.void, 12))).ravel(), np.ascontiguousarray(v).view(np.uint32).view(np.dtype((np.void, 12))).ravel()).all()   # without optimal placement no new position appears
    soup = np.random.default_rng(3).integers(0, 40, (300, 3)).astype(np.uint32)
    xyz, _, tris, _ = oracle.simplify_rounds(np.random.default_rng(4).uniform(0, 1, (40, 3)).astype(np.float32), soup)
    assert tris.max(initial=0) < max(len(xyz), 1) and (tris[:, 0] != tris[:, 1]).all()
