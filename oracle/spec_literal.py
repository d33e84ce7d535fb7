"""oracle/spec_literal.py -- TEST INFRASTRUCTURE ONLY: a second, independent evaluation of the TSDF integrate rule.

The voxel update of SURVEY.md Appendix C, evaluated LITERALLY in numpy float64: plain multiplications, additions and true divisions in
the order the text writes them -- no fused multiply-add, no reciprocal, no packed tricks, nothing shared with oracle/tsdf_oracle.c or
the HIP kernel (which both use the fp32 forms of DESIGN.md 3.5: nested fmaf for the world -> camera transform, fmaf(pc.x*fx, 1/pc.z, mx)
for the projection, fmaf in the running mean).  It answers the question bit-equality between kernel and oracle cannot: how far is the
fp32 kernel from the specification as written?  tests/test_gpu_tsdf.py::test_spec_literal_tolerance measures the north-star tolerance
(1e-4 m on TSDF values) against it on BASELINE configs[1] frames.

    for every voxel of every block that exists at the frame (App. C "integrate (per compacted block, per voxel)"):
        pw = (block*8 + local) * voxel;  pc = T^-1 * pw
        pixel = (int)(pc.x*fx/pc.z + mx + 0.5, pc.y*fy/pc.z + my + 0.5)      # C cast: truncation towards zero
        skip if pc.z <= 0 or the pixel is outside the image
        d = depth[pixel]; skip if -inf or d >= maxDist
        sdf = d - pc.z;  t = trunc(d);  skip if sdf <= -t;  sdf = sdf >= 0 ? min(t, sdf) : max(-t, sdf)
        v.sdf = (v.sdf*v.w + sdf*w_new) / (v.w + w_new);  v.w = min(w_max_eff, v.w + w_new)

Discontinuous steps (the pixel cast, the two skip tests) can legitimately fall on the other side in fp32 when the continuous quantity
sits within rounding distance of the threshold.  `evaluate` reports those voxels (`tie`), so a test can show they are rare and exclude
exactly them -- everything else must agree on the weight and to the tolerance on the sdf.
"""
import numpy as np


def depth_to_metres(depth_u16, depth_shift=1000.0, dmin=0.1, dmax=6.0):
    """App. C depth pre-pass: u16 -> f32 metres; 0, < sensorDepthMin or > sensorDepthMax -> -inf."""
    d = (depth_u16.astype(np.float32) / np.float32(depth_shift)).astype(np.float64)
    bad = (depth_u16 == 0) | (d < dmin) | (d > dmax)
    return np.where(bad, -np.inf, d)


def centre_in_frustum(coords, Ti, *, voxel, fx, fy, mx, my, width, height, dmin, dmax, eps=1e-5):
    """sf_params::frustum_mode 1 (VoxelHashing isSDFBlockInCameraFrustumApprox as remembered from the public sources), literally in
    float64: the block centre ((8 b + 3.5) voxel) to camera space, projected, x / y normalised over (W - 1) / (H - 1) and z over the sensor depth
    range, everything scaled by 0.95 and tested against [-1, 1]^2 x [0, 1].  -> (inside bool [n], near bool [n]: within eps of a boundary)."""
    c = (np.asarray(coords, np.float64) * 8.0 + 3.5) * voxel
    pc = c @ Ti[:3, :3].T + Ti[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = pc[:, 0] * fx / pc[:, 2] + mx
        v = pc[:, 1] * fy / pc[:, 2] + my
        nx = (2.0 * u - (width - 1.0)) / (width - 1.0) * 0.95
        ny = ((height - 1.0) - 2.0 * v) / (height - 1.0) * 0.95
        nz = (pc[:, 2] - dmin) / (dmax - dmin) * 0.95
        inside = (nx >= -1) & (nx <= 1) & (ny >= -1) & (ny <= 1) & (nz >= 0) & (nz <= 1) & (pc[:, 2] > 0)
        near = (np.abs(np.abs(nx) - 1) < eps) | (np.abs(np.abs(ny) - 1) < eps) | (np.abs(nz) < eps) | (np.abs(nz - 1) < eps)
    return inside, near


def evaluate(frames, coords, birth, *, voxel, fx, fy, mx, my, width, height, trunc_base=0.06, trunc_scale=0.02, max_dist=4.0, weight_sample=1,
             weight_max=255, depth_shift=1000.0, dmin=0.1, dmax=6.0, eps_px=None, eps_m=2e-5, frustum_mode=0, weight_mode=0, weight_wrap=0, colour_round=0, colour_first=0, return_colour=False):
    """frames: [(depth u16 [H,W], camToWorld 4x4)], in order.  coords: int [n,3] block coordinates.  birth: int [n] index of the first frame
    at which block i exists (allocation is a separate rule; the caller takes it from the implementation under test).
    frustum_mode 0: every existing block is visited (the sphere test of App. C is conservative: it never keeps a voxel that projects into the
    image from being visited, so it is not evaluated here).  frustum_mode 1: only blocks whose centre passes `centre_in_frustum` for the frame.
    weight_mode 1: an observation weighs (uchar)max(weight_sample * 1.5 * (1 - (d - dmin) / (dmax - dmin)), 1) (VoxelHashing).
    weight_wrap 1: the stored weight is min(weight_max, w + w_new) modulo 256 (upstream's uchar; no clamp at 255).
    A frame may carry a third element, rgb uint8 [H,W,3]: the colour rule of App. C is then evaluated too (first observation copies -- by weight 0, or
    by a black accumulated colour with colour_first 1 --, later ones average per channel, truncating or, with colour_round 1, rounding half up) and
    `return_colour=True` appends the colours (int [n,512,3]) to the result.
    Returns (sdf float64 [n,512], weight int [n,512], tie bool [n,512])."""
    if eps_px is None:
        eps_px = 1e-6 * max(width, height)   # ~8 ulp of a pixel coordinate at the far edge of the image: what a few fp32 roundings can move it
    coords = np.asarray(coords, np.int64)
    n = len(coords)
    l = np.arange(8, dtype=np.int64)
    # voxel index z*64 + y*8 + x
    gx = (coords[:, 0, None, None, None] * 8 + l[None, None, None, :]).astype(np.float64) * voxel
    gy = (coords[:, 1, None, None, None] * 8 + l[None, None, :, None]).astype(np.float64) * voxel
    gz = (coords[:, 2, None, None, None] * 8 + l[None, :, None, None]).astype(np.float64) * voxel
    X = np.broadcast_to(gx, (n, 8, 8, 8)).reshape(n, 512)
    Y = np.broadcast_to(gy, (n, 8, 8, 8)).reshape(n, 512)
    Z = np.broadcast_to(gz, (n, 8, 8, 8)).reshape(n, 512)
    sdf_acc = np.zeros((n, 512), np.float64)
    w_acc = np.zeros((n, 512), np.int64)
    col_acc = np.zeros((n, 512, 3), np.int64)
    tie = np.zeros((n, 512), bool)
    wmax = int(weight_max) if weight_wrap else min(int(weight_max), 255)
    wn = float(weight_sample)
    birth = np.asarray(birth, np.int64)
    for k, frame in enumerate(frames):
        depth, pose = frame[0], frame[1]
        rgb = frame[2] if len(frame) > 2 else None
        pose = np.asarray(pose, np.float64).reshape(4, 4)
        if not np.isfinite(pose).all():
            continue   # tracking lost: the frame is skipped
        Ti = np.linalg.inv(pose)
        df = depth_to_metres(np.asarray(depth), depth_shift, dmin, dmax)
        live = birth <= k
        if frustum_mode == 1:
            inside, near = centre_in_frustum(coords, Ti, voxel=voxel, fx=fx, fy=fy, mx=mx, my=my, width=width, height=height, dmin=dmin, dmax=dmax)
            tie[live & near] = True   # the fp32 test may fall on the other side for the whole block
            live = live & inside
        if not live.any():
            continue
        x, y, z = X[live], Y[live], Z[live]
        pcx = Ti[0, 0] * x + Ti[0, 1] * y + Ti[0, 2] * z + Ti[0, 3]
        pcy = Ti[1, 0] * x + Ti[1, 1] * y + Ti[1, 2] * z + Ti[1, 3]
        pcz = Ti[2, 0] * x + Ti[2, 1] * y + Ti[2, 2] * z + Ti[2, 3]
        front = pcz > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            u = pcx * fx / pcz + mx + 0.5
            v = pcy * fy / pcz + my + 0.5
        pu, pv = np.trunc(u), np.trunc(v)   # the C cast
        inside = front & (pu >= 0) & (pu < width) & (pv >= 0) & (pv < height) & np.isfinite(u) & np.isfinite(v)
        ix = np.where(inside, pu, 0).astype(np.int64)
        iy = np.where(inside, pv, 0).astype(np.int64)
        d = df[iy, ix]
        valid = inside & np.isfinite(d) & (d < max_dist)
        dd = np.where(valid, d, 0.0)
        raw = dd - pcz
        t = trunc_base + trunc_scale * dd
        upd = valid & (raw > -t)
        s = np.where(raw >= 0, np.minimum(t, raw), np.maximum(-t, raw))
        # where a discontinuous step sits within rounding distance of its threshold
        with np.errstate(invalid="ignore"):
            near_u, near_v = np.abs(u - np.rint(u)) < eps_px, np.abs(v - np.rint(v)) < eps_px
        t_here = front & (near_u | near_v) & (u > -1.5) & (u < width + 0.5) & (v > -1.5) & (v < height + 0.5)
        t_here |= np.abs(pcz) < eps_m
        t_here |= valid & (np.abs(raw + t) < eps_m)
        so, wo = sdf_acc[live], w_acc[live]
        if weight_mode == 1:
            z01 = (dd - dmin) / (dmax - dmin)
            wk = np.minimum(np.trunc(np.maximum(weight_sample * 1.5 * (1.0 - z01), 1.0)), 255.0)
            # the (uchar) cast is a step: an observation whose float weight sits within rounding distance of an integer is a tie
            raw_w = weight_sample * 1.5 * (1.0 - z01)
            t_here |= valid & (raw_w > 1.0) & (np.abs(raw_w - np.rint(raw_w)) < 1e-5)
        else:
            wk = np.full(dd.shape, wn)
        if rgb is not None:
            c_new = np.asarray(rgb, np.int64)[iy, ix]                  # [m, 512, 3]
            c_old = col_acc[live]
            first = (c_old.sum(-1) == 0) if colour_first else (wo == 0)
            avg = (c_old + c_new + (1 if colour_round else 0)) // 2
            c_upd = np.where(first[..., None], c_new, avg)
            col_acc[live] = np.where(upd[..., None], c_upd, c_old)
        new_s = (so * wo + s * wk) / (wo + wk)
        sdf_acc[live] = np.where(upd, new_s, so)
        w_new = np.minimum(wmax, wo + wk.astype(np.int64))
        w_acc[live] = np.where(upd, w_new % 256 if weight_wrap else w_new, wo)
        tie[live] |= t_here
    return (sdf_acc, w_acc, tie, col_acc) if return_colour else (sdf_acc, w_acc, tie)
