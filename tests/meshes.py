"""Procedural test meshes + PLY writers for the Segmentator parity tests."""
import struct

import numpy as np


def grid(n=4):
    v = [(x, y, 0) for y in range(n) for x in range(n)]
    f = []
    for y in range(n - 1):
        for x in range(n - 1):
            i = y * n + x
            f += [(i, i + 1, i + n), (i + 1, i + n + 1, i + n)]
    return np.array(v, np.float32), np.array(f, np.uint32)


def bent_strip():
    v = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (2, 0, 1), (2, 1, 1)]
    f = [(0, 1, 2), (1, 3, 2), (1, 4, 3), (4, 5, 3)]
    return np.array(v, np.float32), np.array(f, np.uint32)


def l_shape():
    v, f = grid(4)
    v = [tuple(p) for p in v]
    f = [tuple(t) for t in f]
    for z in range(1, 4):
        for y in range(4):
            v.append((3, y, z))

    def wid(y, z):
        return y * 4 + 3 if z == 0 else 16 + (z - 1) * 4 + y
    for z in range(3):
        for y in range(3):
            a, b, c, d = wid(y, z), wid(y + 1, z), wid(y, z + 1), wid(y + 1, z + 1)
            f += [(a, b, c), (b, d, c)]
    return np.array(v, np.float32), np.array(f, np.uint32)


def bumpy(n=120, seed=5, creases=True):
    """Heightfield with creases, noise, a few degenerate faces and unreferenced vertices."""
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(n, dtype=np.float32) * 0.02, np.arange(n, dtype=np.float32) * 0.02)
    z = 0.05 * np.sin(x * 7) * np.cos(y * 5)
    if creases:
        z = z + 0.3 * (np.abs(x - 1.0) < 0.3) + 0.2 * np.maximum(0, y - 1.5)
    z = z + rng.normal(0, 0.0005, z.shape)
    v = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    f = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            if (i + j) % 2:
                f += [(a, a + 1, a + n), (a + 1, a + n + 1, a + n)]
            else:
                f += [(a, a + 1, a + n + 1), (a, a + n + 1, a + n)]
    f = np.array(f, np.uint32)
    rng.shuffle(f)
    v = np.concatenate([v, rng.normal(0, 1, (7, 3)).astype(np.float32)])  # unreferenced vertices
    return v, f


def write_ply(path, v, f, fmt="ascii", index_name="vertex_indices", colors=False, extra_props=False):
    v = np.asarray(v, np.float32)
    f = np.asarray(f, np.uint32)
    hdr = ["ply", {"ascii": "format ascii 1.0", "le": "format binary_little_endian 1.0", "be": "format binary_big_endian 1.0"}[fmt],
           "comment test mesh", "element vertex %d" % len(v), "property float x", "property float y", "property float z"]
    if extra_props:
        hdr += ["property float nx"]
    if colors:
        hdr += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
    hdr += ["element face %d" % len(f), "property list uchar int %s" % index_name, "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(hdr) + "\n").encode())
        if fmt == "ascii":
            for i, p in enumerate(v):
                s = "%g %g %g" % tuple(p)
                if extra_props:
                    s += " 0.5"
                if colors:
                    s += " %d %d %d 255" % (i % 256, (i * 7) % 256, (i * 13) % 256)
                fh.write((s + "\n").encode())
            for t in f:
                fh.write(("3 %d %d %d\n" % tuple(t)).encode())
        else:
            e = "<" if fmt == "le" else ">"
            for i, p in enumerate(v):
                fh.write(struct.pack(e + "fff", *p))
                if extra_props:
                    fh.write(struct.pack(e + "f", 0.5))
                if colors:
                    fh.write(bytes([i % 256, (i * 7) % 256, (i * 13) % 256, 255]))
            for t in f:
                fh.write(b"\x03" + struct.pack(e + "iii", *[int(k) for k in t]))


def bumpy_large(n=700, seed=11):
    """The same kind of heightfield at scan size (n = 700: 490 007 vertices, 977 202 faces), built without Python loops."""
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(n, dtype=np.float32) * 0.01, np.arange(n, dtype=np.float32) * 0.01)
    z = 0.05 * np.sin(x * 7) * np.cos(y * 5) + 0.3 * (np.abs(x - 2.0) < 0.6) + 0.2 * np.maximum(0, y - 4.5) + 0.15 * (np.hypot(x - 5, y - 2) < 0.8)
    z = z + rng.normal(0, 0.0005, z.shape)
    v = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    j, i = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (j * n + i).reshape(-1).astype(np.uint32)
    odd = ((i + j) % 2).reshape(-1).astype(bool)
    t0 = np.where(odd[:, None], np.stack([a, a + 1, a + n], -1), np.stack([a, a + 1, a + n + 1], -1))
    t1 = np.where(odd[:, None], np.stack([a + 1, a + n + 1, a + n], -1), np.stack([a, a + n + 1, a + n], -1))
    f = np.concatenate([t0, t1]).astype(np.uint32)
    rng.shuffle(f)
    v = np.concatenate([v, rng.normal(0, 1, (7, 3)).astype(np.float32)])  # unreferenced vertices
    return v, f


def write_ply_le_fast(path, v, f, colors=True):
    """Binary little-endian PLY in ScanNet's layout (float x y z, uchar red green blue alpha, list uchar int vertex_indices), vectorised."""
    v = np.asarray(v, np.float32)
    f = np.asarray(f, np.uint32)
    vt = np.dtype([("xyz", "<f4", 3)] + ([("rgba", "u1", 4)] if colors else []))
    va = np.zeros(len(v), vt)
    va["xyz"] = v
    if colors:
        i = np.arange(len(v))
        va["rgba"] = np.stack([i % 256, (i * 7) % 256, (i * 13) % 256, np.full(len(v), 255)], -1).astype(np.uint8)
    fa = np.zeros(len(f), np.dtype([("n", "u1"), ("idx", "<i4", 3)]))
    fa["n"] = 3
    fa["idx"] = f.astype(np.int32)
    hdr = ["ply", "format binary_little_endian 1.0", "element vertex %d" % len(v), "property float x", "property float y", "property float z"]
    if colors:
        hdr += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
    hdr += ["element face %d" % len(f), "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(hdr) + "\n").encode())
        fh.write(va.tobytes())
        fh.write(fa.tobytes())
