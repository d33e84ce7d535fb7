"""Known answers for the quadric edge collapse (the decimator was checked through properties only).

MeshLab / VCG are not in the reference tree and cannot run here, so "the reference's output" is not available; what CAN be pinned is the
published algorithm (Garland & Heckbert quadrics as VCG's TriEdgeCollapseQuadric uses them, parameters from Server/tools/meshclean/
simplify.mlx:3-16) on inputs whose answers follow by hand:

  * the quadric of a vertex = sum over its faces of (n n^T, -2 (n.p) n, (n.p)^2) with n the UN-normalised face normal (area weighting);
  * the collapsed vertex sits at the minimiser of the summed quadric -- on full-rank quadrics the unique minimiser, on rank-deficient ones
    (flat areas, creases) the minimiser CLOSEST TO THE EDGE MIDPOINT (the one documented difference from VCG, which solves the singular
    system with free coordinates at zero and warns of "bad spikes in very flat areas", simplify.mlx:12);
  * priority = ScaleFactor * max(error, 1e-15) / min(QualityThr, worst quality of the surrounding faces after the move), with
    ScaleFactor = 1e8 / diag^6 and quality = 2 area / longest edge^2.

sf_simplify_probe_edge (include/scanfuse_internal.h) returns what the filter computes for one edge before the first collapse."""
import ctypes as C

import numpy as np
import pytest

from scannet_amd import _abi, meshclean
from scannet_amd.segmentator import Mesh


def _probe(xyz, tris, v0, v1, **over):
    L = _abi.lib()
    meshclean._lib()
    p = meshclean.SfSimplifyParams()
    L.sf_simplify_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    L.sf_simplify_probe_edge.argtypes = [C.c_void_p, C.POINTER(meshclean.SfSimplifyParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_float), C.POINTER(C.c_double)]
    m = Mesh.from_arrays(xyz, tris)
    q = np.zeros(10, np.float64)
    pos = np.zeros(3, np.float32)
    pri, sc = C.c_float(0), C.c_double(0)
    _abi.check(L.sf_simplify_probe_edge(m._h, C.byref(p), v0, v1, q.ctypes.data, pos.ctypes.data, C.byref(pri), C.byref(sc)))
    return q, pos, pri.value, sc.value


def _quadric_by_definition(xyz, tris, verts):
    """Sum over the faces around each of `verts` of the plane quadric with the un-normalised normal (closed meshes: no border terms)."""
    A, b, c = np.zeros((3, 3)), np.zeros(3), 0.0
    P = np.asarray(xyz, np.float64)
    for v in verts:
        for t in tris:
            if v in t:
                p0, p1, p2 = P[t[0]], P[t[1]], P[t[2]]
                n = np.cross(p1 - p0, p2 - p0)
                off = n @ p0
                A += np.outer(n, n); b += -2 * off * n; c += off * off
    return A, b, c


def _quality(p0, p1, p2):
    a = np.linalg.norm(np.cross(p1 - p0, p2 - p0))
    return a / max(np.dot(p1 - p0, p1 - p0), np.dot(p2 - p0, p2 - p0), np.dot(p1 - p2, p1 - p2))


CUBE_V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float32) + np.float32([1, 2, 3])
CUBE_T = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.uint32)


def test_quadric_of_a_cube_corner():
    """Vertex 0 of the cube (at p = (1, 2, 3)) lies on the diagonal of its z- and y-faces (two unit-normal triangles each) and off the
    diagonal of its x-face (one): its quadric is (x - 1)^2 + 2 (y - 2)^2 + 2 (z - 3)^2 exactly; the summed quadric of an edge equals the
    definition evaluated in numpy."""
    q, pos, pri, scale = _probe(CUBE_V, CUBE_T, 0, 1)
    A, b, c = _quadric_by_definition(CUBE_V, CUBE_T, [0, 1])
    assert np.allclose([q[0], q[1], q[2], q[3], q[4], q[5]], [A[0, 0], A[0, 1], A[0, 2], A[1, 1], A[1, 2], A[2, 2]], rtol=0, atol=1e-12)
    assert np.allclose(q[6:9], b, atol=1e-12) and abs(q[9] - c) < 1e-12
    A0, b0, c0 = _quadric_by_definition(CUBE_V, CUBE_T, [0])
    p = CUBE_V[0].astype(np.float64)
    D = np.diag([1.0, 2.0, 2.0])
    assert np.allclose(A0, D) and np.allclose(b0, -2 * D @ p) and abs(c0 - p @ D @ p) < 1e-12
    assert abs(scale - 1e8 / 3.0 ** 3) < 1e-3 * scale                                                     # diag^2 = 3
    # the edge 0-1 runs along x: the planes y = 2 and z = 3 hold it, x = 1 (twice, at vertex 0) and x = 2 (vertex 1: once on the diagonal-free
    # faces) pull along it -- the minimiser lies on the edge at the weighted mean of the two x planes
    wx0 = sum(1 for t in CUBE_T if 0 in t and abs(np.cross(CUBE_V[t[1]] - CUBE_V[t[0]], CUBE_V[t[2]] - CUBE_V[t[0]])[0]) > 0.5)
    wx1 = sum(1 for t in CUBE_T if 1 in t and abs(np.cross(CUBE_V[t[1]] - CUBE_V[t[0]], CUBE_V[t[2]] - CUBE_V[t[0]])[0]) > 0.5)
    want_x = (wx0 * 1.0 + wx1 * 2.0) / (wx0 + wx1)
    assert np.allclose(pos, [want_x, 2.0, 3.0], atol=1e-6)
    err = wx0 * (want_x - 1.0) ** 2 + wx1 * (want_x - 2.0) ** 2
    x = pos.astype(np.float64)
    assert abs((x @ A @ x + b @ x + c) - err) < 1e-9
    # priority by the definition: the faces around 0 and 1 that do not hold the other end, with that end moved to x
    quals = []
    for a, other in ((0, 1), (1, 0)):
        for t in CUBE_T:
            if a in t and other not in t:
                pts = [pos.astype(np.float64) if k == a else CUBE_V[k].astype(np.float64) for k in t]
                quals.append(_quality(*pts))
    want = scale * err / min(0.3, min(quals))
    assert abs(pri - want) < 1e-5 * want


def _grid(n, z=None):
    xs = np.arange(n, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    Z = np.zeros_like(X) if z is None else z(X, Y).astype(np.float32)
    v = np.stack([X, Y, Z], -1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    return v, np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.uint32)


def test_coplanar_edge_costs_the_floor_and_stays_on_the_midpoint():
    """An interior edge of a flat grid: the quadric error is 0 everywhere on the plane, so the collapse costs the floor 1e-15 (QuadricEpsilon)
    over the quality term, and the rank-1 quadric leaves the position free in the plane: the minimiser closest to the midpoint IS the midpoint."""
    v, t = _grid(7)
    a, b = 3 * 7 + 3, 3 * 7 + 4
    q, pos, pri, scale = _probe(v, t, a, b)
    assert np.allclose(pos, 0.5 * (v[a] + v[b]), atol=1e-6) and pos[2] == 0.0
    A = np.array([[q[0], q[1], q[2]], [q[1], q[3], q[4]], [q[2], q[4], q[5]]])
    assert np.linalg.matrix_rank(A, tol=1e-9) == 1 and A[2, 2] > 0
    quals = []
    for p_, other in ((a, b), (b, a)):
        for tri in t:
            if p_ in tri and other not in tri:
                pts = [pos.astype(np.float64) if k == p_ else v[k].astype(np.float64) for k in tri]
                quals.append(_quality(*pts))
    want = np.float32(1e-15 / min(0.3, min(quals)))
    assert pri == pytest.approx(float(want), rel=1e-6)
    assert pri < 1e-14      # far below any collapse that bends the surface: flat areas go first


def test_crease_edge_lands_on_the_crease():
    """A sheet folded along x = 3 (z rises behind the fold): an edge from a crease vertex to a vertex on the flat side carries two planes --
    rank 2.  The minimisers form the line where both planes' errors vanish, the crease itself; the one closest to the edge midpoint is the
    midpoint's projection onto it."""
    v, t = _grid(7, z=lambda X, Y: np.maximum(X - 3, 0) * 0.75)
    a, b = 3 * 7 + 3, 3 * 7 + 2          # (x 3, y 3) on the crease, (x 2, y 3) on the flat side
    q, pos, pri, scale = _probe(v, t, a, b)
    A = np.array([[q[0], q[1], q[2]], [q[1], q[3], q[4]], [q[2], q[4], q[5]]])
    assert np.linalg.matrix_rank(A, tol=1e-9) == 2
    # both planes vanish exactly on the crease line {x = 3, z = 0}; the point of it closest to the midpoint (2.5, 3, 0) is (3, 3, 0)
    assert np.allclose(pos, [3.0, 3.0, 0.0], atol=1e-5)
    x = pos.astype(np.float64)
    assert abs(x @ A @ x + q[6:9] @ x + q[9]) < 1e-9
    # an edge ALONG the crease keeps its midpoint
    c = 4 * 7 + 3
    _, pos2, _, _ = _probe(v, t, a, c)
    assert np.allclose(pos2, [3.0, 3.5, 0.0], atol=1e-5)


def test_full_rank_minimiser_is_the_plane_intersection():
    """Three planes in general position meet in one point: a vertex whose faces lie in them (a skewed corner) and any edge at it collapse to
    exactly that point, error 0 -- the closed form of the adjugate-inverse branch."""
    apex = np.array([0.3, -0.2, 0.9])
    d = [np.array([1.0, 0.1, -0.3]), np.array([-0.2, 1.0, -0.4]), np.array([-0.5, -0.6, -0.8])]
    v = np.array([apex, apex + d[0], apex + d[1], apex + d[2]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]], np.uint32)     # a tetrahedron: closed, no border terms
    q, pos, pri, scale = _probe(v, t, 0, 1)
    A, b, c = _quadric_by_definition(v, t, [0, 1])
    want = np.linalg.solve(A, -0.5 * b)
    assert np.allclose(pos, want, atol=1e-5)
    # vertex 0 alone: its three faces meet in the apex
    A0, b0, c0 = _quadric_by_definition(v, t, [0])
    assert np.allclose(np.linalg.solve(A0, -0.5 * b0), v[0].astype(np.float64), atol=1e-6)


def test_sequential_filter_removes_the_flat_part_first():
    """End to end on a known answer: a flat 21 x 21 grid with nine isolated pyramids.  Collapsing to the face budget that the flat part alone
    can supply must leave every pyramid apex where it was (their collapses cost orders of magnitude more than the floor)."""
    n = 21
    apex = [(5, 5), (5, 10), (5, 15), (10, 5), (10, 10), (10, 15), (15, 5), (15, 10), (15, 15)]
    v, t = _grid(n, z=lambda X, Y: sum(((X == ax) & (Y == ay)) * 1.0 for ax, ay in apex))
    out, st = meshclean.simplify(Mesh.from_arrays(v, t), target_perc=0.5)
    xyz = out.arrays()[0]
    assert st["faces_out"] <= 0.5 * len(t) + 2
    for ax, ay in apex:
        assert np.min(np.linalg.norm(xyz - np.float32([ax, ay, 1.0]), axis=1)) < 1e-6, (ax, ay)
    assert st["max_priority"] < 1e-6      # nothing but floor-cost collapses was needed


@pytest.mark.gpu
def test_gpu_filter_removes_the_flat_part_first():
    """The same known answer through the rounds of independent collapses on the GPU (sf_mesh_simplify_gpu): every pyramid apex stays."""
    n = 41
    apex = [(x, y) for x in (8, 16, 24, 32) for y in (8, 16, 24, 32)]
    v, t = _grid(n, z=lambda X, Y: sum(((X == ax) & (Y == ay)) * 1.0 for ax, ay in apex))
    out, st = meshclean.simplify(Mesh.from_arrays(v, t), gpu=0, target_perc=0.5)
    xyz = out.arrays()[0]
    assert st["faces_out"] <= 0.5 * len(t) + 2
    for ax, ay in apex:
        assert np.min(np.linalg.norm(xyz - np.float32([ax, ay, 1.0]), axis=1)) < 1e-6, (ax, ay)
