"""sf_mesh_merge_parts (the C++ merge bin/depthsensing --ranks N ends with) against partition.merge_slab_meshes (the numpy rule the multi-GPU tests and
bench.py's prefix check use: tests/test_shard_gloo.py, tests/test_multi_gpu.py) on synthetic parts: keyed vertices with shared boundary keys,
interleaved cube keys, an empty part, a part without faces, no colour.  No GPU: both sides are host code."""
import numpy as np
import pytest

from scannet_amd import partition
from scannet_amd.segmentator import Mesh


def _parts(rng, world, nv_total, nf_per, shared=0.1, interleave=True, empty=None, no_faces=None):
    """A keyed 'mesh' cut into `world` parts: vertex keys are distinct random u64, a fraction `shared` of them present in two parts (same position,
    same colour, as an edge on a stripe boundary has); every part's vertices ascend by key and its faces by cube key, as marching cubes gives them."""
    keys = np.unique(rng.integers(1, 1 << 62, nv_total * 2, dtype=np.uint64))[:nv_total]
    rng.shuffle(keys)
    xyz = rng.standard_normal((nv_total, 3)).astype(np.float32)
    rgba = rng.integers(0, 256, (nv_total, 4), dtype=np.uint8)
    owner = rng.integers(0, world, nv_total)
    second = np.where(rng.random(nv_total) < shared, (owner + 1) % world, -1)
    parts = []
    for r in range(world):
        mine = np.flatnonzero((owner == r) | (second == r))
        if r == empty:
            mine = mine[:0]
        mine = mine[np.argsort(keys[mine])]
        nf = 0 if (r == no_faces or len(mine) < 3) else nf_per
        tris = rng.integers(0, max(len(mine), 1), (nf, 3)).astype(np.uint32)
        fk = np.sort(rng.integers(0, 1 << 40, nf, dtype=np.uint64) * (world if interleave else 1) + (r if interleave else (r << 50)))
        parts.append((xyz[mine], rgba[mine], tris, keys[mine], fk))
    return parts


@pytest.mark.parametrize("world,kw", [(2, {}), (3, {"empty": 1}), (4, {"no_faces": 2}), (2, {"shared": 0.0}), (5, {"shared": 0.5}), (1, {})])
def test_merge_parts_is_the_numpy_rule(world, kw):
    rng = np.random.default_rng(world * 7 + len(kw))
    parts = _parts(rng, world, 5000, 3000, **kw)
    want = partition.merge_slab_meshes(parts)
    meshes = [Mesh.from_arrays(p[0], p[2], rgba=p[1], keys=p[3], face_keys=p[4]) for p in parts]
    m = Mesh.merge_parts(meshes)
    xyz, rgba, tris, keys = m.arrays(keys=True)
    assert np.array_equal(keys, want[3]) and np.array_equal(xyz.view(np.uint32), want[0].view(np.uint32)) and np.array_equal(rgba, want[1])
    assert np.array_equal(tris, want[2])
    fk = m.face_keys()
    assert len(fk) == len(tris) and np.all(fk[1:] >= fk[:-1])          # the merged mesh carries its keys: merges nest
    again = Mesh.merge_parts([m])
    assert all(np.array_equal(a, b) for a, b in zip(again.arrays(keys=True), (xyz, rgba, tris, keys)))


@pytest.mark.parametrize("interleave", [True, False])
def test_merge_at_a_size_that_takes_several_threads(interleave):
    """Above ~200 k elements per thread the merge cuts the key space into ranges and merges them side by side (csrc/ply.cpp split_runs): shared keys
    on both sides of a cut, faces interleaved key by key (worst case) or in long runs (what stripes give)."""
    rng = np.random.default_rng(3)
    parts = _parts(rng, 3, 500000, 400000, shared=0.3, interleave=interleave)
    if not interleave:   # long runs: stripes of cube keys dealt round-robin
        parts = [(p[0], p[1], p[2], p[3], np.sort(((np.arange(len(p[4]), dtype=np.uint64) // 5000) * 3 + r) * 100000 + (p[4] % 100000))) for r, p in enumerate(parts)]
    want = partition.merge_slab_meshes(parts)
    m = Mesh.merge_parts([Mesh.from_arrays(p[0], p[2], rgba=p[1], keys=p[3], face_keys=p[4]) for p in parts])
    xyz, rgba, tris, keys = m.arrays(keys=True)
    assert np.array_equal(keys, want[3]) and np.array_equal(xyz, want[0]) and np.array_equal(rgba, want[1]) and np.array_equal(tris, want[2])


def test_merge_of_parts_that_are_not_sorted():
    """Marching cubes hands sorted parts over (the k-way merge); arrays in any other order go through the sort and give the same mesh."""
    rng = np.random.default_rng(23)
    parts = []
    for xyz, rgba, tris, keys, fk in _parts(rng, 3, 4000, 2500, shared=0.2):
        perm = rng.permutation(len(keys))                      # new position -> old vertex
        inv = np.argsort(perm)
        fperm = rng.permutation(len(fk))
        parts.append((xyz[perm], rgba[perm], inv[tris.astype(np.int64)].astype(np.uint32)[fperm], keys[perm], fk[fperm]))
    want = partition.merge_slab_meshes(parts)
    m = Mesh.merge_parts([Mesh.from_arrays(p[0], p[2], rgba=p[1], keys=p[3], face_keys=p[4]) for p in parts])
    xyz, rgba, tris, keys = m.arrays(keys=True)
    assert np.array_equal(keys, want[3]) and np.array_equal(xyz, want[0]) and np.array_equal(rgba, want[1])
    # equal cube keys within one shuffled part have no defined order in either rule beyond stability: compare as sorted-by-key groups
    fk = m.face_keys()
    assert np.all(fk[1:] >= fk[:-1]) and np.array_equal(tris, want[2])


def test_merge_without_face_keys_keeps_the_parts_in_order():
    """Contiguous slabs: no face keys needed, faces stay part after part (merge_slab_meshes with 4-tuples)."""
    rng = np.random.default_rng(5)
    parts = [p[:4] for p in _parts(rng, 3, 2000, 900, interleave=False)]
    want = partition.merge_slab_meshes(parts)
    m = Mesh.merge_parts([Mesh.from_arrays(p[0], p[2], rgba=p[1], keys=p[3]) for p in parts])
    xyz, rgba, tris, keys = m.arrays(keys=True)
    assert np.array_equal(keys, want[3]) and np.array_equal(xyz, want[0]) and np.array_equal(rgba, want[1]) and np.array_equal(tris, want[2])
    with pytest.raises(Exception):
        m.face_keys()


def test_merge_of_nothing_and_of_unkeyed_meshes():
    m = Mesh.merge_parts([])
    assert m.counts() == (0, 0)
    plain = Mesh.from_arrays(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32))
    with pytest.raises(Exception, match="no vertex keys"):
        Mesh.merge_parts([plain])
    with pytest.raises(ValueError):
        Mesh.from_arrays(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32), keys=np.zeros(2, np.uint64))


def test_merged_ply_is_the_single_mesh_ply(tmp_path):
    """What the tool writes: the merged mesh's PLY equals the PLY of the same mesh built in one piece."""
    rng = np.random.default_rng(11)
    parts = _parts(rng, 2, 3000, 2500)
    want = partition.merge_slab_meshes(parts)
    a, b = str(tmp_path / "merged.ply"), str(tmp_path / "whole.ply")
    Mesh.merge_parts([Mesh.from_arrays(p[0], p[2], rgba=p[1], keys=p[3], face_keys=p[4]) for p in parts]).write_ply(a)
    Mesh.from_arrays(want[0], want[2], rgba=want[1]).write_ply(b)
    assert open(a, "rb").read() == open(b, "rb").read()
