"""sf_mesh_write_ply against the file format spelled out in numpy (README.md:45-46 of the reference: binary little-endian PLY, vertex float x y z +
uchar red green blue alpha, face list uchar int vertex_indices): header, 16-byte vertex records, 13-byte face records -- at sizes where one thread
writes the file and where several threads format and pwrite slices of it (csrc/ply.cpp), with and without colour, and read back by sf_ply_read."""
import numpy as np
import pytest

from scannet_amd.segmentator import Mesh

HEADER = ("ply\nformat binary_little_endian 1.0\ncomment scanfuse-mi355x\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
          "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n")


@pytest.mark.parametrize("nv,nf,colour", [(0, 0, True), (5, 3, True), (5, 0, False), (262144, 262145, True), (700001, 1300003, False), (1300003, 700001, True)])
def test_ply_bytes(tmp_path, nv, nf, colour):
    rng = np.random.default_rng(nv + nf)
    xyz = rng.standard_normal((nv, 3)).astype(np.float32)
    tris = rng.integers(0, max(nv, 1), (nf, 3)).astype(np.uint32)
    rgba = rng.integers(0, 256, (nv, 4), dtype=np.uint8) if colour else None
    path = str(tmp_path / "m.ply")
    Mesh.from_arrays(xyz, tris, rgba=rgba).write_ply(path)
    v = np.zeros(nv, dtype=[("p", "<f4", 3), ("c", "u1", 4)])
    v["p"] = xyz
    v["c"] = rgba if colour else 255          # a mesh without colour is written white and opaque
    f = np.zeros(nf, dtype=[("n", "u1"), ("i", "<u4", 3)])
    f["n"] = 3
    f["i"] = tris
    assert open(path, "rb").read() == (HEADER % (nv, nf)).encode() + v.tobytes() + f.tobytes()
    back = Mesh.read(path).arrays()
    assert np.array_equal(back[0].view(np.uint32), xyz.view(np.uint32)) and np.array_equal(back[2], tris)
    assert np.array_equal(back[1], rgba if colour else np.full((nv, 4), 255, np.uint8))


def test_ply_write_failure_is_reported(tmp_path):
    m = Mesh.from_arrays(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32))
    with pytest.raises(Exception, match="unable to open"):
        m.write_ply(str(tmp_path / "no" / "such" / "dir" / "m.ply"))


def test_ply_into_a_pipe(tmp_path):
    """A destination that cannot seek (a FIFO, /dev/stdout) gets the same bytes from one thread writing in file order."""
    import os
    import subprocess
    rng = np.random.default_rng(1)
    nv, nf = 600001, 700003
    m = Mesh.from_arrays(rng.standard_normal((nv, 3)).astype(np.float32), rng.integers(0, nv, (nf, 3)).astype(np.uint32), rgba=rng.integers(0, 256, (nv, 4), dtype=np.uint8))
    plain, fifo, got = str(tmp_path / "file.ply"), str(tmp_path / "fifo.ply"), str(tmp_path / "from_fifo.ply")
    m.write_ply(plain)
    os.mkfifo(fifo)
    with open(got, "wb") as out:
        p = subprocess.Popen(["cat", fifo], stdout=out)
        m.write_ply(fifo)
        assert p.wait(timeout=60) == 0
    assert open(plain, "rb").read() == open(got, "rb").read()
