"""Mesh cleaning: the C++ filters of scannet_amd/csrc/clean.cpp against the independent numpy/scipy
restatement oracle/clean_oracle.py, known answers, the shipped filter scripts, and the meshlabserver-compatible CLI.
PARITY UNPINNED against MeshLab itself (not available; no version pinned by the reference) -- see the oracle header."""
import os
import subprocess

import numpy as np
import pytest

from scannet_amd import meshclean
from scannet_amd.segmentator import Mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MLX = "/root/reference/Server/tools/meshclean"
CLEAN_MLX = """<!DOCTYPE FilterScript>
<FilterScript>
 <filter name="Merge Close Vertices">
  <Param min="0" max="0.106888" description="Merging distance" name="Threshold" tooltip="x" value="0.0010689" type="RichAbsPerc"/>
 </filter>
 <filter name="Remove Duplicate Faces"/>
 <filter name="Remove Isolated pieces (wrt Face Num.)">
  <Param description="Enter minimum conn. comp size:" name="MinComponentSize" tooltip="x" value="%d" type="RichInt"/>
 </filter>
 <filter name="Remove Unreferenced Vertex"/>
</FilterScript>
"""


def _grid(nx, ny, origin=(0, 0, 0), step=0.01):
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny))
    xyz = np.stack([xs.ravel() * step + origin[0], ys.ravel() * step + origin[1], np.full(xs.size, origin[2])], 1).astype(np.float32)
    tris = []
    for y in range(ny - 1):
        for x in range(nx - 1):
            i = y * nx + x
            tris += [(i, i + 1, i + nx), (i + 1, i + nx + 1, i + nx)]
    return xyz, np.array(tris, np.uint32)


def _soup(rng):
    """Two sheets + a small island + noise: near-duplicate vertices (triangle-soup seams), duplicate and flipped faces,
    unreferenced vertices."""
    a_xyz, a_tri = _grid(30, 30)
    b_xyz, b_tri = _grid(12, 12, origin=(1.0, 0, 0))
    c_xyz, c_tri = _grid(3, 3, origin=(2.0, 0, 0))
    xyz = np.concatenate([a_xyz, b_xyz, c_xyz])
    tris = np.concatenate([a_tri, b_tri + len(a_xyz), c_tri + len(a_xyz) + len(b_xyz)])
    # un-weld 300 random corners: a copy of the vertex displaced by < threshold
    extra = []
    tris = tris.copy()
    for k in rng.choice(tris.size, 300, replace=False):
        f, c = divmod(int(k), 3)
        v = tris[f, c]
        d = rng.normal(size=3)
        d *= rng.uniform(0, 0.0009) / np.linalg.norm(d)
        extra.append(xyz[v] + d.astype(np.float32))
        tris[f, c] = len(xyz) + len(extra) - 1
    # a chain of vertices 0.0008 apart: greedy clustering is NOT transitive
    chain0 = len(xyz) + len(extra)
    chain = np.array([[3.0 + 0.0008 * i, 0, 0] for i in range(7)], np.float32)
    far = np.array([[3.0, 0.5, 0], [3.0, 0.5, 0.4]], np.float32)
    xyz = np.concatenate([xyz, np.array(extra, np.float32), chain, far, rng.uniform(5, 6, (10, 3)).astype(np.float32)])
    chain_tris = np.array([(chain0 + i, chain0 + 7, chain0 + 8) for i in range(7)], np.uint32)
    dup = np.concatenate([tris[5:25], tris[40:50][:, ::-1]])
    tris = np.concatenate([tris, chain_tris, dup]).astype(np.uint32)
    rgba = rng.integers(0, 256, (len(xyz), 4), dtype=np.uint8)
    return xyz, rgba, tris


@pytest.mark.parametrize("seed,min_cc", [(0, 100), (1, 10), (2, 2000), (3, 1)])
def test_against_independent_restatement(seed, min_cc):
    from oracle import clean_oracle
    xyz, rgba, tris = _soup(np.random.default_rng(seed))
    m = Mesh.from_arrays(xyz, tris, rgba)
    out, st = meshclean.clean(m, meshclean.CLEAN_MLX_MERGE_DISTANCE, min_cc)
    gx, gc, gt = out.arrays()
    ox, oc, ot = clean_oracle.clean(xyz, rgba, tris, meshclean.CLEAN_MLX_MERGE_DISTANCE, min_cc)
    assert np.array_equal(gx.view(np.uint32), ox.view(np.uint32))
    assert np.array_equal(gc, oc)
    assert np.array_equal(gt, ot)
    assert st["vertices_out"] == len(ox) and st["faces_out"] == len(ot)
    assert st["faces_duplicate"] >= 30 and st["vertices_merged"] >= 300


def test_known_answers():
    # greedy, non-transitive clustering on a chain with 0.0008 spacing and radius 0.0010689: centres 0, 2, 4, 6
    chain = np.array([[0.0008 * i, 0, 0] for i in range(7)], np.float32)
    apex = np.array([[0, 1, 0], [0, 1, 1]], np.float32)
    xyz = np.concatenate([chain, apex])
    tris = np.array([(i, 7, 8) for i in range(7)], np.uint32)
    out, st = meshclean.clean(Mesh.from_arrays(xyz, tris), 0.0010689, 0)
    gx, _, gt = out.arrays()
    assert np.array_equal(gx, xyz[[0, 2, 4, 6, 7, 8]])
    assert gt.tolist() == [[0, 4, 5], [1, 4, 5], [2, 4, 5], [3, 4, 5]]   # merged faces became duplicates of their centre's face
    assert st["vertices_merged"] == 3 and st["faces_duplicate"] == 3 and st["faces_degenerate"] == 0
    # strict '<': a vertex at exactly the threshold distance is not merged
    r = np.float32(0.0010689)
    xyz = np.array([[0, 0, 0], [r, 0, 0], [0, 1, 0], [0, 1, 1]], np.float32)
    out, st = meshclean.clean(Mesh.from_arrays(xyz, np.array([(0, 2, 3), (1, 2, 3)], np.uint32)), float(r), 0)
    assert st["vertices_merged"] == 0 and st["faces_out"] == 2
    # component size: 'fewer than' -- a component with exactly min faces survives
    xyz, tris = _grid(3, 3)  # 8 faces
    for min_cc, keep in ((8, 8), (9, 0)):
        out, st = meshclean.clean(Mesh.from_arrays(xyz, tris), 0.0, min_cc)
        assert st["faces_out"] == keep and st["vertices_out"] == (9 if keep else 0)
    # empty mesh
    out, st = meshclean.clean(Mesh.from_arrays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32)))
    assert out.counts() == (0, 0)


def test_shipped_scripts_and_cli(tmp_path):
    for name, want in (("clean.mlx", 7500), ("cleanLoRes.mlx", 1000)):
        path = os.path.join(REF_MLX, name)
        if not os.path.exists(path):  # the GPU box has no /root/reference: same text, written here
            path = str(tmp_path / name)
            open(path, "w").write(CLEAN_MLX % want)
        s = meshclean.load_script(path)
        assert (s.merge_close_vertices, s.remove_duplicate_faces, s.remove_small_components, s.remove_unreferenced) == (1, 1, 1, 1)
        assert abs(s.merge_distance - 0.0010689) < 1e-9 and s.min_component_faces == want
    unknown = tmp_path / "unknown.mlx"
    unknown.write_text('<!DOCTYPE FilterScript>\n<FilterScript>\n <filter name="Laplacian Smooth">\n </filter>\n</FilterScript>\n')
    with pytest.raises(Exception, match="not implemented"):
        meshclean.load_script(str(unknown))
    # CLI with meshlabserver's flags, in place (-i == -o, scan_processor.py:134)
    xyz, rgba, tris = _soup(np.random.default_rng(5))
    ply = str(tmp_path / "scene_vh.ply")
    Mesh.from_arrays(xyz, tris, rgba).write_ply(ply)
    script = tmp_path / "c.mlx"
    script.write_text(CLEAN_MLX % 100)
    exe = os.path.join(ROOT, "bin", "meshclean")
    r = subprocess.run([exe, "-i", ply, "-o", ply, "-m", "vc", "-s", str(script)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "" and "Mesh saved as" in r.stdout
    from oracle import clean_oracle
    ox, oc, ot = clean_oracle.clean(xyz, rgba, tris, 0.0010689, 100)
    gx, gc, gt = Mesh.read(ply).arrays()
    assert np.array_equal(gx, ox) and np.array_equal(gc, oc) and np.array_equal(gt, ot)
    bad = subprocess.run([exe, "-i", str(tmp_path / "nope.ply"), "-o", ply, "-m", "vc", "-s", str(script)], capture_output=True, text=True)
    assert bad.returncode != 0 and bad.stderr != ""
