"""TEST INFRASTRUCTURE -- an INDEPENDENT evaluation of the marching-cubes stage of SURVEY.md Appendix C in float64, written
from the text and sharing no code with oracle/mc_oracle.c or scannet_amd/csrc/mc.hip: which grid edges carry a vertex, where the vertex
sits, what colour it has, and how many triangles there are.  The case table only enters through its triangle COUNT per case (the table
itself is verified from first principles by tests/test_mc_tables.py); vertex sets, positions and colours do not depend on it.

    evaluate(coords, vox, voxel, thresh_factor, ntri_per_case) -> dict(keys u64[nv] ascending, pos f64[nv,3], col f64[nv,3] (unrounded), n_tris)

A cube (8 voxels g + {0,1}^3) takes part iff all 8 exist, have weight > 0 and |sdf| <= thresh_factor * voxel; corner i is inside iff sdf < 0;
an edge of such a cube whose end points differ in sign carries a vertex at mu = s_lo / (s_lo - s_hi) from its lower end point, colour
c_lo + mu * (c_hi - c_lo).  parity unpinned against the external DepthSensing binary, like the rest of the TSDF stage.
"""
import numpy as np

KEY_BIAS = 1 << 19
CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], np.int64)
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def _pack(g):
    return ((g[..., 0] + KEY_BIAS) << 42) | ((g[..., 1] + KEY_BIAS) << 22) | ((g[..., 2] + KEY_BIAS) << 2)


def evaluate(coords, vox, voxel, thresh_factor, ntri_per_case):
    coords = np.asarray(coords, np.int64)
    l = np.arange(512)
    local = np.stack([l & 7, (l >> 3) & 7, l >> 6], 1)                      # index z*64 + y*8 + x
    g = (coords[:, None, :] * 8 + local[None, :, :]).reshape(-1, 3)
    sdf = vox["sdf"].reshape(-1).astype(np.float64)
    w = vox["w"].reshape(-1)
    rgb = np.stack([vox["r"].reshape(-1), vox["g"].reshape(-1), vox["b"].reshape(-1)], 1).astype(np.float64)
    thresh = float(np.float32(thresh_factor) * np.float32(voxel))           # the parameter product as the float the files hold
    good = (w > 0) & (np.abs(sdf) <= thresh)
    keys = _pack(g)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]

    def lookup(q):
        """index into the voxel arrays of grid points q [m,3], -1 where no such voxel exists"""
        k = _pack(q)
        pos = np.searchsorted(skeys, k)
        pos[pos >= len(skeys)] = 0
        hit = skeys[pos] == k
        return np.where(hit, order[pos], -1)

    corner = np.stack([lookup(g + CORNERS[i]) for i in range(8)], 1)        # [N, 8]
    ok = (corner >= 0).all(1)
    ok &= good[np.where(corner >= 0, corner, 0)].all(1)
    cube = np.nonzero(ok)[0]
    cidx = corner[cube]                                                      # [M, 8]
    s = sdf[cidx]
    inside = s < 0.0
    case = (inside * (1 << np.arange(8))).sum(1)
    n_tris = int(np.asarray(ntri_per_case)[case].sum())
    vk, vp, vc = [], [], []
    for a, c in EDGES:
        axis = int(np.nonzero(CORNERS[a] != CORNERS[c])[0][0])
        if CORNERS[a][axis] > CORNERS[c][axis]:
            a, c = c, a
        cut = inside[:, a] != inside[:, c]
        if not cut.any():
            continue
        lo, hi = cidx[cut, a], cidx[cut, c]
        gl = g[lo]
        mu = sdf[lo] / (sdf[lo] - sdf[hi])
        p = gl.astype(np.float64)
        p[:, axis] += mu
        vk.append(_pack(gl) | axis)
        vp.append(p * float(np.float32(voxel)))
        vc.append(rgb[lo] + mu[:, None] * (rgb[hi] - rgb[lo]))
    if not vk:
        return dict(keys=np.zeros(0, np.uint64), pos=np.zeros((0, 3)), col=np.zeros((0, 3)), n_tris=0)
    vk, vp, vc = np.concatenate(vk), np.concatenate(vp), np.concatenate(vc)
    uk, first = np.unique(vk, return_index=True)                            # an edge is shared by up to four cubes: same end points, same vertex
    return dict(keys=uk.astype(np.uint64), pos=vp[first], col=vc[first], n_tris=n_tris)
