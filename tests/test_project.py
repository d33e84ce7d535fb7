"""Annotation projection (second half): AnnotationTools/ProjectAnnotations -- the labelled mesh drawn into every
frame, the depth-consistency and 5x5 filters, and the vertex labelling in front of them.
CPU part: the checker oracle/project_oracle.c against closed-form geometry (pixel alignment, ray/plane hits through the near-plane
clip, depth test), the host logic (aggregation / segs / label map -> vertex ids, propagation against a brute-force search), a golden
digest.  GPU part (-m gpu): scannet_amd/csrc/project.hip against the checker, bit for bit.  PARITY UNPINNED against the reference
binary (Direct3D 11 + mLib + FreeImage, no test or golden image in the tree): see the checker's header for the conventions fixed here."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from scannet_amd import project

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_PATH = os.path.join(ROOT, "tests", "golden", "project_golden.json")
CW, CH, DW, DH = 162, 121, 80, 60
FX = FY = 146.0


def _params(cw=CW, ch=CH, dw=DW, dh=DH, fx=FX, fy=FY, **kw):
    return project.default_params((cw, ch), (dw, dh), fx, fy, **kw)


def _pose(eye, yaw, pitch=0.0):
    """camera-to-world, vision convention (x right, y down, z forward), world z up"""
    f = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch)])
    r = np.cross(f, [0.0, 0.0, 1.0])
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    T = np.eye(4, dtype=np.float32)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = r, d, f, eye
    return T


def _quad(p0, p1, p2, p3):
    return [p0, p1, p2, p3], [[0, 1, 2], [0, 2, 3]]


def _grid_patch(origin, du, dv, nu, nv):
    """(nu x nv) quads over origin + a du + b dv -> vertices, triangles"""
    o, du, dv = np.asarray(origin, float), np.asarray(du, float), np.asarray(dv, float)
    a, b = np.mgrid[0:nu + 1, 0:nv + 1]
    v = o + a[..., None] * du / nu + b[..., None] * dv / nv
    idx = lambda i, j: i * (nv + 1) + j
    t = []
    for i in range(nu):
        for j in range(nv):
            t += [[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)], [idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)]]
    return v.reshape(-1, 3), np.array(t)


def room_scene(n=10, seed=0):
    """A 6 x 4 x 3 m box room from tessellated patches, two boxes on the floor, per-vertex ids by patch (with a few unlabelled
    vertices), plus a sprinkle of tiny and sliver triangles."""
    rng = np.random.default_rng(seed)
    patches = [([0, 0, 0], [6, 0, 0], [0, 4, 0]), ([0, 0, 3], [0, 4, 0], [6, 0, 0]), ([0, 0, 0], [0, 0, 3], [6, 0, 0]), ([0, 4, 0], [6, 0, 0], [0, 0, 3]),
               ([0, 0, 0], [0, 4, 0], [0, 0, 3]), ([6, 0, 0], [0, 0, 3], [0, 4, 0]),
               ([2, 1.5, 0.8], [1, 0, 0], [0, 1, 0]), ([2, 1.5, 0], [0, 0, 0.8], [1, 0, 0]), ([3, 1.5, 0], [0, 0, 0.8], [0, 1, 0]),
               ([4.2, 2.6, 0.45], [0.6, 0, 0], [0, 0.5, 0])]
    V, T, I, Lb = [], [], [], []
    base = 0
    for k, (o, du, dv) in enumerate(patches):
        v, t = _grid_patch(o, du, dv, n, n)
        v = v + rng.normal(0, 0.002, v.shape)
        V.append(v); T.append(t + base)
        I.append(np.full(len(v), k + 1)); Lb.append(np.full(len(v), 100 * (k % 4) + 7 + k))
        base += len(v)
    m = 60
    c = rng.uniform([0.5, 0.5, 0.2], [5.5, 3.5, 2.5], (m, 3))
    tiny = c[:, None, :] + rng.normal(0, 0.01, (m, 3, 3)) * np.array([1, 1, 1])
    tiny[::3, 2] = tiny[::3, 0] + (tiny[::3, 1] - tiny[::3, 0]) * 1.0001   # slivers
    V.append(tiny.reshape(-1, 3)); T.append(np.arange(3 * m).reshape(m, 3) + base)
    I.append(np.full(3 * m, 40)); Lb.append(np.full(3 * m, 4000))
    xyz = np.concatenate(V).astype(np.float32)
    tris = np.concatenate(T).astype(np.uint32)
    inst = np.concatenate(I).astype(np.uint8)
    label = np.concatenate(Lb).astype(np.uint16)
    drop = rng.random(len(xyz)) < 0.03
    inst[drop] = 0
    label[drop] = 0
    return xyz, tris, inst, label


def room_poses():
    return np.stack([_pose([3.0, 2.0, 1.5], 0.3), _pose([1.0, 1.0, 1.2], 1.2, -0.4), _pose([5.0, 3.0, 0.4], 3.6, 0.2), _pose([2.6, 2.0, 1.0], -0.2, -0.9),
                     _pose([0.2, 0.2, 0.15], 0.8, 0.05), _pose([3.0, 2.0, 2.9], 2.0, -1.2)])


def sensor_depth(params, xyz, tris, inst, label, poses, seed=1):
    """a plausible sensor depth per pose: the checker's own rendered depth, resampled, with noise, holes and one gross outlier region"""
    rng = np.random.default_rng(seed)
    out = []
    for T in poses:
        _, _, z = orc.project_frame(params, xyz, tris, inst, label, T, None, want_depth=True)
        ys = np.round(np.arange(params.depth_height) * (params.color_height - 1) / (params.depth_height - 1)).astype(int)
        xs = np.round(np.arange(params.depth_width) * (params.color_width - 1) / (params.depth_width - 1)).astype(int)
        d = (z[np.ix_(ys, xs)] * 1000.0 + rng.integers(-15, 16, (params.depth_height, params.depth_width))).clip(0, 65535)
        d[rng.random(d.shape) < 0.03] = 0
        # the reference's tolerance is depthDistThresh + 0.01f * dorig with dorig in MILLIMETRES (Visualizer.cpp:157), i.e. 0.2 m + 10 x the
        # sensor depth: only a sensor reading far nearer than the mesh removes a label
        d[: params.depth_height // 3, : params.depth_width // 4] = 25
        out.append(d.astype(np.uint16))
    return np.stack(out)


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------------------------- checker, closed forms
def test_pixel_alignment_and_fill_rule():
    """a fronto-parallel square whose edges project exactly onto pixel boundaries covers exactly those pixels"""
    P = _params(cw=64, ch=48, dw=32, dh=24, fx=64.0, fy=64.0)
    z = 2.0
    x0, x1, y0, y1 = [(px - 32) * z / 64.0 for px in (10, 20)] + [(py - 24) * z / 64.0 for py in (5, 9)]
    v, t = _quad([x0, y0, z], [x1, y0, z], [x1, y1, z], [x0, y1, z])
    inst, label, zc = orc.project_frame(P, v, t, [3] * 4, [700] * 4, np.eye(4), None, want_depth=True)
    want = np.zeros((48, 64), bool)
    want[5:9, 10:20] = True
    # the 5x5 vote cannot remove anything here: every pixel of a 10 x 4 block sees >= 20 % of its own label
    assert np.array_equal(label != 0, want) and np.array_equal(inst[want], np.full(want.sum(), 3)) and set(label[want]) == {700}
    assert np.allclose(zc[want], z, atol=2e-4) and np.all(zc[~want] == 0)


def test_flat_colour_comes_from_the_first_vertex():
    P = _params(cw=64, ch=48, dw=32, dh=24, fx=64.0, fy=64.0)
    v = [[-2, -2, 2.0], [2, -2, 2.0], [2, 2, 2.0], [-2, 2, 2.0]]
    t = [[1, 2, 0], [3, 0, 2]]                      # first vertices: 1 and 3
    inst, label = orc.project_frame(P, v, t, [1, 2, 3, 4], [10, 20, 30, 40], np.eye(4))
    assert set(np.unique(label)) == {20, 40} and set(np.unique(inst)) == {2, 4}
    yy, xx = np.mgrid[0:48, 0:64]
    upper = (xx + 0.5 - 32) / 64 - (yy + 0.5 - 24) / 64 > 0.05    # well on the (1, 2, 0) side of the diagonal x = y
    assert np.all(label[upper] == 20) and np.all(label[~upper & ((xx + 0.5 - 32) / 64 - (yy + 0.5 - 24) / 64 < -0.05)] == 40)


def test_depth_test_and_draw_order():
    P = _params(cw=64, ch=48, dw=32, dh=24, fx=64.0, fy=64.0)
    far, _ = _quad([-3, -3, 3.0], [3, -3, 3.0], [3, 3, 3.0], [-3, 3, 3.0])
    near, _ = _quad([-0.5, -0.5, 1.5], [0.5, -0.5, 1.5], [0.5, 0.5, 1.5], [-0.5, 0.5, 1.5])
    same = [list(q) for q in far]                                                # the same two triangles again, drawn later: LESS keeps the first
    v = np.array(far + near + same)
    t = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11]])
    vi = [1] * 4 + [2] * 4 + [3] * 4
    inst, label, zc = orc.project_frame(P, v, t, vi, [x * 10 for x in vi], np.eye(4), None, want_depth=True)
    assert set(np.unique(inst)) == {1, 2}
    assert np.all(inst[24 - 10:24 + 10, 32 - 10:32 + 10] == 2) and np.allclose(zc[24, 32], 1.5, atol=1e-4) and np.allclose(zc[2, 2], 3.0, atol=1e-3)


def test_near_plane_clip_against_ray_plane_hits():
    """a huge floor passing under and behind the camera: every pixel whose ray meets it between depth_min and depth_max is labelled"""
    P = _params()
    T = _pose([0.0, 0.0, 0.12], 0.4, -0.05)
    v, t = _quad([-60, -60, 0], [60, -60, 0], [60, 60, 0], [-60, 60, 0])
    inst, label, zc = orc.project_frame(P, v, t, [9] * 4, [99] * 4, T, None, want_depth=True)
    yy, xx = np.mgrid[0:CH, 0:CW]
    ray = np.stack([(xx + 0.5 - CW / 2) / FX, (yy + 0.5 - CH / 2) / FY, np.ones_like(xx, float)], -1) @ T[:3, :3].T.astype(float)
    with np.errstate(divide="ignore", invalid="ignore"):
        tz = np.where(ray[..., 2] < 0, -0.12 / ray[..., 2], np.inf)       # camera z of the hit (the ray's camera z component is 1)
    sure_in, sure_out = (tz > 0.1 * 1.02) & (tz < 15.0 * 0.98), (tz < 0.1 * 0.98) | (tz > 15.0 * 1.02)
    assert sure_in.sum() > 3000 and (tz[sure_in] < 0.5).sum() > 300          # the clipped part is really in view
    assert np.all(label[sure_in] == 99) and np.all(label[sure_out] == 0)
    assert np.allclose(zc[sure_in], tz[sure_in], rtol=2e-3)


def test_depth_consistency_and_vote_filters():
    P = _params()
    xyz, tris, inst, label = room_scene()
    T = room_poses()[0]
    i0, l0, z = orc.project_frame(P, xyz, tris, inst, label, T, None, want_depth=True)
    ys = np.round(np.arange(DH) * (CH - 1) / (DH - 1)).astype(int)
    xs = np.round(np.arange(DW) * (CW - 1) / (DW - 1)).astype(int)
    d = (z[np.ix_(ys, xs)] * 1000).astype(np.uint16)
    i1, l1 = orc.project_frame(P, xyz, tris, inst, label, T, d)
    assert np.array_equal(l0, l1) and np.array_equal(i0, i1)                 # the sensor agrees with the mesh: nothing removed
    d2 = d.copy()
    d2[:, : DW // 2] = 20                                                    # sensor sees something at 2 cm on the left half: tolerance 0.4 m
    i2, l2 = orc.project_frame(P, xyz, tris, inst, label, T, d2)
    left = np.zeros((CH, CW), bool)
    left[:, : CW // 2 - 3] = True
    big_gap = left & (z > 0.5)
    assert np.all(l2[big_gap] == 0) and np.array_equal(l2[:, CW // 2 + 3:], l0[:, CW // 2 + 3:])
    d3 = d.copy()
    d3[10:20, 10:30] = 0
    i3, l3 = orc.project_frame(P, xyz, tris, inst, label, T, d3)
    assert np.array_equal(l3, l0)                                            # holes in the sensor depth are ignored by default ...
    Pf = _params(filter_using_original_depth=1)
    i4, l4 = orc.project_frame(Pf, xyz, tris, inst, label, T, d3)
    assert np.all(l4[24:38, 24:56] == 0) and (l4 != 0).sum() > 0.5 * (l0 != 0).sum()      # ... unless asked for
    # vote: a 1-pixel speck of a foreign label inside a uniform wall disappears, a 3 x 3 patch (9/25) stays
    Pv = _params(cw=64, ch=48, dw=32, dh=24, fx=64.0, fy=64.0)
    wall, wt = _quad([-3, -3, 3.0], [3, -3, 3.0], [3, 3, 3.0], [-3, 3, 3.0])
    px = lambda a, b, c, e, zz: _quad(*[[(x - 32) * zz / 64, (y - 24) * zz / 64, zz] for x, y in ((a, c), (b, c), (b, e), (a, e))])[0]
    v = np.array(wall + px(10, 11, 10, 11, 2.0) + px(40, 43, 30, 33, 2.0))
    t = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11]])
    vi = [1] * 4 + [2] * 4 + [3] * 4
    iv, lv = orc.project_frame(Pv, v, t, vi, [x * 10 for x in vi], np.eye(4))
    assert lv[10, 10] == 0 and np.all(lv[30:33, 40:43] == 30) and lv[20, 20] == 10


def test_invalid_pose_gives_empty_images():
    P = _params()
    xyz, tris, inst, label = room_scene(4)
    T = np.full((4, 4), -np.inf, np.float32)
    i, l = orc.project_frame(P, xyz, tris, inst, label, T)
    assert not i.any() and not l.any()


def test_matrix_is_the_rigid_inverse_times_projection():
    T = _pose([1.0, 2.0, 0.5], 0.7, -0.3)
    M = orc.project_matrix(T, 500.0, 510.0, 640, 480, 0.1, 15.0).astype(np.float64)
    p_cam = np.array([0.3, -0.2, 2.5, 1.0])
    clip = M @ (T.astype(np.float64) @ p_cam)
    assert np.allclose(clip[3], 2.5, atol=1e-5)
    assert np.allclose((clip[0] / clip[3] + 1) * 320, 320 + 500.0 * 0.3 / 2.5, atol=1e-3)
    assert np.allclose((1 - clip[1] / clip[3]) * 240, 240 + 510.0 * -0.2 / 2.5, atol=1e-3)
    assert np.allclose(clip[2] / clip[3], 15.0 / 14.9 * (1 - 0.1 / 2.5), atol=1e-6)


def test_golden_digest():
    P = _params()
    xyz, tris, inst, label = room_scene()
    poses = room_poses()
    depth = sensor_depth(P, xyz, tris, inst, label, poses)
    digs = []
    for k, T in enumerate(poses):
        i, l, z = orc.project_frame(P, xyz, tris, inst, label, T, depth[k], want_depth=True)
        assert (l != 0).mean() > 0.3
        digs.append(_sha(i, l, z))
    if os.environ.get("SF_WRITE_GOLDEN"):
        json.dump({"scene": "room_scene(10, 0) x room_poses(), sensor_depth seed 1, 162x121 / 80x60, fx 146", "sha256": digs}, open(GOLDEN_PATH, "w"), indent=1)
    assert json.load(open(GOLDEN_PATH))["sha256"] == digs


# ---------------------------------------------------------------------------------------------------------------- host logic
def _write_scene_files(tmp_path, seg, groups, tsv_rows):
    segs = tmp_path / "s_vh_clean_2.0.010000.segs.json"
    segs.write_text(json.dumps({"params": {"kThresh": "0.01", "segMinVerts": "20"}, "sceneId": "s", "segIndices": [int(x) for x in seg]}))
    agg = tmp_path / "s.aggregation.json"
    agg.write_text(json.dumps({"sceneId": "s", "appId": "Aggregator.v2", "segGroups": groups, "segmentsFile": "s_vh_clean_2.0.010000.segs.json"}, indent=2))
    tsv = tmp_path / "labels.tsv"
    tsv.write_text("id\tcategory\tcount\n" + "".join("%d\t%s\t1\n" % (k, c) for k, c in enumerate(tsv_rows)))
    return str(segs), str(agg), str(tsv)


def test_vertex_ids_from_aggregation(tmp_path):
    rng = np.random.default_rng(2)
    seg = rng.integers(0, 12, 500) * 7 + 3
    segids = sorted(set(seg))
    groups = [{"id": 0, "objectId": 0, "segments": [int(segids[0]), int(segids[1])], "label": "chair"},
              {"id": 1, "objectId": 1, "segments": [int(segids[2])], "label": "a \"quoted\" table"},
              {"id": 2, "objectId": 2, "segments": [int(segids[3]), int(segids[4])], "label": "chair"},       # same category as object 0
              {"id": 3, "objectId": 3, "segments": [int(segids[5])], "label": "not in the map"},
              {"id": 4, "objectId": 4, "segments": [int(segids[6]), 99999], "label": "lamp"}]
    rows = ["wall", "", "chair", "lamp", "a \"quoted\" table"]      # ids = 1-based line numbers; the empty category does not get an id but counts as a line
    paths = _write_scene_files(tmp_path, seg, groups, rows)
    inst, label, nlab = project.vertex_ids(*paths, len(seg))
    ids = {"wall": 1, "chair": 3, "lamp": 4, "a \"quoted\" table": 5}
    want_i, want_l = np.zeros(500, np.uint8), np.zeros(500, np.uint16)
    first = {}
    for k, g in enumerate(groups):
        if g["label"] not in ids:
            continue
        first.setdefault(ids[g["label"]], k + 1)                    # the colour table is keyed by label: first object's index + 1
        for s in g["segments"]:
            want_i[seg == s] = first[ids[g["label"]]]
            want_l[seg == s] = ids[g["label"]]
    assert nlab == 3 and np.array_equal(inst, want_i) and np.array_equal(label, want_l)
    assert set(inst[seg == segids[3]]) == {1}                        # object 2 ("chair") carries object 0's instance value
    with pytest.raises(Exception):
        project.vertex_ids(*paths, len(seg) + 1)
    with pytest.raises(Exception):
        project.vertex_ids(paths[0], str(tmp_path / "missing.json"), paths[2], len(seg))


def test_propagation_against_brute_force():
    rng = np.random.default_rng(5)
    sv, st = _grid_patch([0, 0, 0], [4, 0, 0], [0, 3, 0], 24, 18)
    sv = sv + rng.normal(0, 0.01, sv.shape)
    sv[:, 2] += 0.3 * np.sin(sv[:, 0] * 2)
    dv, dt = _grid_patch([-0.2, -0.2, 0], [4.4, 0, 0], [0, 3.4, 0], 60, 45)
    dv[:, 2] += 0.3 * np.sin(dv[:, 0] * 2) + rng.normal(0, 0.02, len(dv))
    # a second sheet facing the other way just above the first: normals must decide
    dv2 = dv.copy(); dv2[:, 2] += 0.05
    dt2 = dt[:, ::-1] + len(dv)
    dv, dt = np.concatenate([dv, dv2]), np.concatenate([dt, dt2])
    si = (1 + (sv[:, 0] > 2) + 2 * (sv[:, 1] > 1.5)).astype(np.uint8)
    sl = (si * 11).astype(np.uint16)
    unl = rng.random(len(sv)) < 0.2
    si[unl] = 0; sl[unl] = 0
    di, dl = project.propagate(sv, st, si, sl, dv, dt, 0.5)

    def normals(v, t):
        v = v.astype(np.float32); n = np.zeros_like(v)
        c = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
        for k in range(3):
            np.add.at(n, t[:, k], c)
        return n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    ns, nd = normals(sv, st), normals(dv, dt)
    ext = (sv.astype(np.float32).max(0) - sv.astype(np.float32).min(0)).max()
    thr = max(ext * 0.01, 0.05)
    idx = np.nonzero(sl > 0)[0]
    wi, wl = np.zeros(len(dv), np.uint8), np.zeros(len(dv), np.uint16)
    S = sv.astype(np.float32)[idx]
    ambiguous = 0
    for i, p in enumerate(dv.astype(np.float32)):
        d2 = ((S - p) ** 2).sum(1)
        o = np.lexsort((np.arange(len(d2)), d2))[:3]
        if np.any(np.abs(d2[o] - thr) < 1e-5) or (len(d2) > 3 and abs(np.sort(d2)[3] - np.sort(d2)[2]) < 1e-7):
            ambiguous += 1; wi[i], wl[i] = di[i], dl[i]; continue
        all_same, hit = True, -1
        for k in range(3):
            if d2[o[k]] < thr:
                ang = np.arccos(np.clip(float(ns[idx[o[k]]] @ nd[i]), -1, 1))
                if abs(ang - 0.5) < 1e-4:
                    ambiguous += 1; hit = -2; break
                if ang < 0.5:
                    hit = k; break
                if si[idx[o[k]]] != si[idx[o[0]]]:
                    all_same = False
            else:
                all_same = False
        if hit == -2:
            wi[i], wl[i] = di[i], dl[i]
        elif hit >= 0:
            wi[i], wl[i] = si[idx[o[hit]]], sl[idx[o[hit]]]
        elif all_same:
            wi[i], wl[i] = si[idx[o[0]]], sl[idx[o[0]]]
    assert ambiguous < 0.01 * len(dv)
    assert np.array_equal(di, wi) and np.array_equal(dl, wl)
    n1 = len(dv) // 2
    assert (dl[:n1] != 0).mean() > 0.6          # the sheet that faces like the source takes its ids ...
    inside = (dv[n1:, 0] > 0.3) & (dv[n1:, 0] < 1.7) & (dv[n1:, 1] > 0.3) & (dv[n1:, 1] < 1.2)
    flipped = di[n1:][inside]                   # ... the flipped one only where three neighbours are in range and agree (one region: instance 1)
    assert set(np.unique(flipped)) <= {0, 1} and (flipped == 1).mean() > 0.7


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_matches_checker_bit_for_bit():
    P = _params()
    xyz, tris, inst, label = room_scene()
    poses = room_poses()
    depth = sensor_depth(P, xyz, tris, inst, label, poses)
    bad = np.full((1, 4, 4), -np.inf, np.float32)
    batch = np.concatenate([poses[:3], bad, poses[3:]])
    dbatch = np.concatenate([depth[:3], depth[:1], depth[3:]])
    golden = json.load(open(GOLDEN_PATH))["sha256"]
    with project.Projector(P) as pr:
        pr.set_mesh(xyz, tris, inst, label)
        gi, gl, gz, us = pr.run(batch, dbatch, want_depth=True)
        assert not gi[3].any() and not gl[3].any()
        keep = [0, 1, 2, 4, 5, 6]
        for k, b in enumerate(keep):
            oi, ol, oz = orc.project_frame(P, xyz, tris, inst, label, poses[k], depth[k], want_depth=True)
            assert np.array_equal(gz[b], oz), "rendered depth, pose %d" % k
            assert np.array_equal(gi[b], oi) and np.array_equal(gl[b], ol), "ids, pose %d" % k
            assert _sha(gi[b], gl[b], gz[b]) == golden[k]
        # no sensor depth: the consistency filter is skipped; one frame at a time equals the batch
        gi1, gl1, _ = pr.run(poses[1:2], None)
        oi, ol = orc.project_frame(P, xyz, tris, inst, label, poses[1], None)
        assert np.array_equal(gi1[0], oi) and np.array_equal(gl1[0], ol)
        # a second mesh on the same projector
        xyz2, tris2, inst2, label2 = room_scene(5, seed=3)
        pr.set_mesh(xyz2, tris2, inst2, label2)
        gi2, gl2, _ = pr.run(poses[:2], depth[:2])
        for k in range(2):
            oi, ol = orc.project_frame(P, xyz2, tris2, inst2, label2, poses[k], depth[k])
            assert np.array_equal(gi2[k], oi) and np.array_equal(gl2[k], ol)


@pytest.mark.gpu
def test_gpu_full_size_frames_and_triangle_order():
    """1296 x 968 over 640 x 480 (ScanNet's sizes): a finer room against the checker on two poses; the result does not depend on the
    order the triangles are stored in, apart from exact depth ties (none here: every triangle gets the same ids as its patch)"""
    P = _params(cw=1296, ch=968, dw=640, dh=480, fx=1170.0, fy=1170.0)
    xyz, tris, inst, label = room_scene(48, seed=2)
    poses = room_poses()[[0, 4]]
    depth = sensor_depth(P, xyz, tris, inst, label, poses)
    with project.Projector(P) as pr:
        pr.set_mesh(xyz, tris, inst, label)
        gi, gl, gz, us = pr.run(poses, depth, want_depth=True)
        for k in range(2):
            oi, ol, oz = orc.project_frame(P, xyz, tris, inst, label, poses[k], depth[k], want_depth=True)
            assert np.array_equal(gz[k], oz) and np.array_equal(gi[k], oi) and np.array_equal(gl[k], ol)
            assert (gl[k] != 0).mean() > 0.3
        # reversed storage order with the first vertex kept: same zcam everywhere, same ids wherever depths do not tie exactly
        pr.set_mesh(xyz, tris[::-1], inst, label)
        ri, rl, rz, _ = pr.run(poses, depth, want_depth=True)
        assert np.array_equal(rz, gz)
        assert (rl != gl).mean() < 1e-3


@pytest.mark.gpu
def test_gpu_argument_checks():
    P = _params()
    with project.Projector(P) as pr:
        with pytest.raises(Exception):
            pr.run(room_poses()[:1])                         # no mesh yet
        xyz, tris, inst, label = room_scene(3)
        pr.set_mesh(xyz, tris, inst, label)
        with pytest.raises(Exception):
            pr.run(np.tile(np.eye(4, dtype=np.float32), (pr.max_batch + 1, 1, 1)))
    with pytest.raises(Exception):
        project.Projector(_params(fx=0.0))


@pytest.mark.gpu
def test_tool_end_to_end(tmp_path):
    """bin/projectannotations on a small scan directory (.sens + decimated and hi-res PLY + segs + aggregation + label map + meta file):
    PNGs out, every frame compared with the checker fed with the same vertex ids; frame skip; invalid pose; missing inputs."""
    import subprocess
    from scannet_amd import filter2d, sens
    from scannet_amd.segmentator import Mesh
    scan = tmp_path / "scans" / "scene7"
    scan.mkdir(parents=True)
    lo_xyz, lo_tris, _, _ = room_scene(6, seed=4)
    hi_xyz, hi_tris, _, _ = room_scene(14, seed=4)
    Mesh.from_arrays(lo_xyz, lo_tris).write_ply(str(scan / "scene7_vh_clean_2.ply"))
    Mesh.from_arrays(hi_xyz, hi_tris).write_ply(str(scan / "scene7_vh_clean.ply"))
    # segments: 0.75 m cells of the decimated mesh; objects: groups of cells
    cell = (np.floor(lo_xyz[:, 0] / 0.75) + 8 * np.floor(lo_xyz[:, 1] / 0.75) + 64 * np.floor(lo_xyz[:, 2] / 0.75)).astype(int)
    cats = ["floor", "wall", "chair", "table", "lamp"]
    groups = [{"id": k, "objectId": k, "segments": [int(s) for s in sorted(set(cell)) if s % 7 == k], "label": cats[k % 5]} for k in range(6)]
    segs, agg, tsv = _write_scene_files(scan, cell, groups, ["wall", "floor", "chair", "", "table"])      # lamp is not in the map
    os.rename(segs, scan / "scene7_vh_clean_2.0.010000.segs.json")
    os.rename(agg, scan / "scene7.aggregation.json")
    P = _params()
    poses = room_poses()
    vi, vl, _ = project.vertex_ids(str(scan / "scene7_vh_clean_2.0.010000.segs.json"), str(scan / "scene7.aggregation.json"), tsv, len(lo_xyz))
    hi_i, hi_l = project.propagate(lo_xyz, lo_tris, vi, vl, hi_xyz, hi_tris, 0.5)
    assert (hi_l != 0).mean() > 0.4
    depth = sensor_depth(P, hi_xyz, hi_tris, hi_i, hi_l, poses)
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = FX, FY, CW / 2, CH / 2
    sd = sens.SensorData.create(CW, CH, DW, DH, K, K, color_compression=0, depth_compression=1)
    for k in range(len(poses)):
        sd.add_frame(depth[k], poses[k] if k != 2 else np.full((4, 4), -np.inf, np.float32), color=np.zeros((CH, CW, 3), np.uint8))
    sd.save(str(scan / "scene7.sens"))
    sd.close()
    (scan / "scene7.txt").write_text("colorWidth = %d\ncolorHeight = %d\ndepthWidth = %d\ndepthHeight = %d\n" % (CW, CH, DW, DH))
    out = tmp_path / "out"
    par = tmp_path / "zParametersScan.txt"

    def write_params(hi, skip):
        par.write_text('s_scanDir = "nowhere/";\ns_outDir = "%s";\ns_outputDebugImages = false;\ns_labelMappingFile = "%s";\n'
                       's_useHiResMesh = %s;\t//comment\ns_filterUsingOrigialDepthImage = false\ns_frameSkip = %d;\ns_depthMin = 0.1f;\n'
                       's_depthMax = 15.0f;\ns_depthDistThresh = 0.2f;\ns_propagateNormalThresh = 0.5f;\n' % (out, tsv, "true" if hi else "false", skip))
    exe = os.path.join(ROOT, "bin", "projectannotations")
    write_params(True, 1)
    r = subprocess.run([exe, str(par), str(scan) + "/"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "", r.stderr + r.stdout
    assert "[ProjectAnnotations]" in r.stdout and "done" in r.stdout
    covered = []
    for k in range(len(poses)):
        gi = filter2d.png_read(str(out / "scene7" / "instance" / ("%d.png" % k)))
        gl = filter2d.png_read(str(out / "scene7" / "label" / ("%d.png" % k)))
        assert gi.dtype == np.uint8 and gl.dtype == np.uint16
        if k == 2:
            assert not gi.any() and not gl.any()
            continue
        oi, ol = orc.project_frame(P, hi_xyz, hi_tris, hi_i, hi_l, poses[k], depth[k])
        assert np.array_equal(gi, oi) and np.array_equal(gl, ol), k
        covered.append((gl != 0).mean())
    assert np.mean(covered) > 0.1
    # decimated mesh only, every second frame
    import shutil
    shutil.rmtree(out)
    write_params(False, 2)
    r = subprocess.run([exe, str(par), str(scan)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    assert sorted(os.listdir(out / "scene7" / "label")) == ["0.png", "2.png", "4.png"]
    oi, ol = orc.project_frame(P, lo_xyz, lo_tris, vi, vl, poses[4], depth[4])
    assert np.array_equal(filter2d.png_read(str(out / "scene7" / "label" / "4.png")), ol)
    assert np.array_equal(filter2d.png_read(str(out / "scene7" / "instance" / "4.png")), oi)
    # missing inputs: a warning on stdout, nothing written, rc 0 (Visualizer.cpp:21-25); no meta file: ERROR (main.cpp:44-47)
    os.rename(scan / "scene7.aggregation.json", scan / "hidden.json")
    r = subprocess.run([exe, str(par), str(scan)], capture_output=True, text=True)
    assert r.returncode == 0 and "WARNING: no sens/mesh/segs/aggregation file, skipping" in r.stdout
    os.remove(scan / "scene7.txt")
    r = subprocess.run([exe, str(par), str(scan)], capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR: meta-file" in r.stdout
