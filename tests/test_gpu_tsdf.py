"""Parity of the HIP fusion path (through the C ABI) against the CPU oracle: bit-exact block sets and voxels.

All tests here need an MI355X (`-m gpu`).  Inputs are seeded / closed-form (scannet_amd/synth.py); the
oracle is oracle/tsdf_oracle.c ("parity unpinned" vs upstream -- see its header -- but pinned analytically in
tests/test_oracle_tsdf.py).
"""
import numpy as np
import pytest

from scannet_amd import synth

pytestmark = pytest.mark.gpu


def _mk(oracle, W=640, H=480, voxel=0.004, **over):
    from scannet_amd import fusion
    op = oracle.default_params(W, H, voxel)
    fx, fy, mx, my = synth.intrinsics(W, H)
    op.fx, op.fy, op.mx, op.my = fx, fy, mx, my
    gp = fusion.default_params(depth_width=W, depth_height=H, voxel_size=voxel, fx=fx, fy=fy, mx=mx, my=my,
                               num_sdf_blocks=over.pop("num_sdf_blocks", 1 << 18))
    for k, v in over.items():
        setattr(gp, k, v)
        if hasattr(op, k):
            setattr(op, k, v)
    return op, gp


def _assert_same(ovol, fuser):
    oc, ov = ovol.export()
    gc, gv = fuser.export_blocks()
    assert len(oc) == len(gc), "block count differs: oracle %d gpu %d" % (len(oc), len(gc))
    assert np.array_equal(oc, gc), "allocated block sets differ"
    same = ov.view(np.uint8).reshape(len(oc), -1) == gv.view(np.uint8).reshape(len(gc), -1)
    if not same.all():
        bad = np.argwhere(~same.reshape(len(oc), 512, 8).all(-1))
        b, v = bad[0]
        raise AssertionError("%d voxels differ; first: block %s voxel %d oracle %s gpu %s" %
                             (len(bad), oc[b], v, ov[b, v], gv[b, v]))


def test_plane_bit_exact(oracle):
    from scannet_amd import fusion
    op, gp = _mk(oracle)
    ovol = oracle.Volume(op, threads=8)
    I = np.eye(4, dtype=np.float32)
    d = synth.plane_frame()
    n = ovol.integrate(d, I)
    with fusion.Fuser(gp) as f:
        assert f.integrate(d, I)
        st = f.stats()
        assert st["last_frame_blocks"] == n
        assert st["blocks_allocated"] == ovol.num_blocks
        assert st["alloc_failures"] == 0
        _assert_same(ovol, f)


def test_room_stream_bit_exact(oracle):
    """Config-2 style walk (noise on): 6 frames spread over the loop, compared after every frame count."""
    from scannet_amd import fusion
    op, gp = _mk(oracle, num_sdf_blocks=1 << 19)
    ovol = oracle.Volume(op, threads=8)
    with fusion.Fuser(gp) as f:
        for i in (0, 1, 2, 700, 1400, 1401):
            pose = synth.trajectory_pose(i, 5578)
            d = synth.render_room_depth(pose, noise_frame=i)
            n = ovol.integrate(d, pose)
            assert f.integrate(d, pose)
            assert f.stats()["last_frame_blocks"] == n
        _assert_same(ovol, f)


@pytest.mark.parametrize("tail_wide", [1, 0])
def test_batched_pass_equals_frame_by_frame(oracle, tail_wide):
    """Temporal blocking: sf_fuser_integrate_batch_device fuses up to 32 frames per pass over the tiles.  The result must
    be the frame-by-frame result bit for bit: blocks born in the middle of a batch (large pose jumps) only receive the
    frames from their birth on, skipped poses leave gaps, the final batch is partial.  The call's LAST pass runs the variant of k_integrate
    that fuses the tile in halves at 8 waves per SIMD (tail_wide 1, the default) or the same kernel as every other pass (0): the same voxels."""
    from scannet_amd import fusion
    W, H = 320, 240
    op, gp = _mk(oracle, W, H, voxel=0.008, num_sdf_blocks=1 << 17)
    ovol = oracle.Volume(op, threads=8)
    idx = [0, 1, 2, 300, 301, 3, 600, 601, 900, 4, 5, 1100, 302, 303, 6, 7, 8, 602, 9, 901, 902, 10, 11,
           12, 13, 304, 305, 14, 603, 604, 15, 16, 903, 17, 18, 1101, 19, 306, 20, 21, 605, 22, 23, 904, 24, 25, 26]  # 47 frames: passes of 8 (ramp), 16 and 23
    depth = np.zeros((len(idx), H, W), np.uint16)
    poses = np.zeros((len(idx), 16), np.float32)
    last = None
    for k, i in enumerate(idx):
        pose = synth.trajectory_pose(i, 1200)
        depth[k] = synth.render_room_depth(pose, W, H, noise_frame=i)
        if k in (5, 17):
            pose = np.full((4, 4), -np.inf, np.float32)  # tracking lost
        else:
            last = ovol.integrate(depth[k], pose)
        poses[k] = pose.reshape(16)
    import ctypes as C
    from scannet_amd import _abi
    L = _abi.lib()
    dptr = C.c_void_p()
    _abi.check(L.sf_device_malloc(0, depth.nbytes, C.byref(dptr)))
    try:
        _abi.check(L.sf_device_upload(dptr, depth.ctypes.data_as(C.c_void_p), depth.nbytes))
        with fusion.Fuser(gp, tail_wide=tail_wide) as f:
            assert f.batch_frames == 32
            f.integrate_batch_device(dptr.value, W * H * 2, poses)
            st = f.stats()
            assert st["frames_integrated"] == len(idx) - 2 and st["frames_skipped"] == 2
            assert st["last_frame_blocks"] == last
            _assert_same(ovol, f)
    finally:
        L.sf_device_free(dptr)


@pytest.mark.parametrize("batch", [2, 3, 4])
def test_passes_of_a_few_frames_with_births_in_the_middle(oracle, batch):
    """Passes of 2, 3 and 4 frames (tune batch): the compaction of such a pass is k_compactify_few's loop over the frames with the birth-frame mask -- eight directory entries
    per thread, one list-counter atomic per 2 048 -- not the lane-per-(entry, frame) kernel of longer passes.  Pose jumps put blocks' births in the middle of a pass, skipped poses
    leave gaps: the oracle's volume, bit for bit, and the last frame's block count."""
    from scannet_amd import fusion
    W, H = 320, 240
    op, gp = _mk(oracle, W, H, voxel=0.008, num_sdf_blocks=1 << 17)
    ovol = oracle.Volume(op, threads=8)
    idx = [0, 1, 300, 301, 2, 600, 3, 4, 900, 901, 5, 302, 6, 7, 601, 8, 9, 10, 1100, 11, 303, 12, 13]
    depth = np.zeros((len(idx), H, W), np.uint16)
    poses = np.zeros((len(idx), 16), np.float32)
    last = None
    for k, i in enumerate(idx):
        pose = synth.trajectory_pose(i, 1200)
        depth[k] = synth.render_room_depth(pose, W, H, noise_frame=i)
        if k in (4, 15):
            pose = np.full((4, 4), -np.inf, np.float32)  # tracking lost
        else:
            last = ovol.integrate(depth[k], pose)
        poses[k] = pose.reshape(16)
    dev = _DeviceFrames(depth)
    try:
        with fusion.Fuser(gp, batch=batch) as f:
            dev.fuse(f, poses, 0, len(idx))
            st = f.stats()
            assert st["frames_integrated"] == len(idx) - 2 and st["frames_skipped"] == 2 and st["alloc_failures"] == 0
            assert st["last_frame_blocks"] == last
            _assert_same(ovol, f)
    finally:
        dev.close()


def test_colour_deintegrate_gc_ragged(oracle):
    """Ragged image size (not a multiple of 8 or 16), colour fusion, deintegration and garbage collection."""
    from scannet_amd import fusion
    W, H = 161, 123
    op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 16)
    ovol = oracle.Volume(op, threads=4)
    rng = np.random.default_rng(7)
    with fusion.Fuser(gp) as f:
        frames = []
        for i in (0, 30, 60):
            pose = synth.trajectory_pose(i, 400)
            d = synth.render_room_depth(pose, W, H, noise_frame=i)
            d[rng.random(d.shape) < 0.05] = 0  # holes
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            frames.append((d, pose, rgb))
            ovol.integrate(d, pose, rgb=rgb)
            assert f.integrate(d, pose, rgb=rgb)
        _assert_same(ovol, f)
        d, pose, rgb = frames[1]
        ovol.deintegrate(d, pose, rgb=rgb)
        assert f.deintegrate(d, pose, rgb=rgb)
        _assert_same(ovol, f)
        freed_o = ovol.garbage_collect()
        freed_g = f.garbage_collect()
        assert freed_o == freed_g and freed_o > 0
        _assert_same(ovol, f)
        # the freed heap slots are reusable: fuse another frame
        pose = synth.trajectory_pose(90, 400)
        d = synth.render_room_depth(pose, W, H)
        ovol.integrate(d, pose)
        assert f.integrate(d, pose)
        _assert_same(ovol, f)
        assert f.stats()["alloc_failures"] == 0


def test_garbage_collection_inside_a_long_stream(oracle):
    """s_garbageCollectionEnabled = true (zParametersScanNet.txt:83): collection runs repeatedly INSIDE a stream.  A walk whose far walls
    come and go: blocks observed only beyond the truncation band are freed every few frames (most of each frame is deintegrated again to
    make that happen often), later frames re-allocate them.  After every collection the hash table holds exactly the live blocks (no
    tombstone survives: probe chains do not grow over a long scan), nothing fails to allocate, and the volume stays bit-identical to the
    oracle's -- one frame per pass and 16 frames per pass."""
    import torch
    from scannet_amd import fusion
    W, H = 160, 120
    op, gp = _mk(oracle, W, H, voxel=0.02, num_sdf_blocks=1 << 14, hash_num_buckets=1 << 11)   # a small table: chains would show
    ovol = oracle.Volume(op, threads=8)
    frames = [(synth.render_room_depth(synth.trajectory_pose(7 * i, 600), W, H, noise_frame=i), synth.trajectory_pose(7 * i, 600)) for i in range(96)]
    with fusion.Fuser(gp) as f, fusion.Fuser(gp) as g:
        freed_total = 0
        for a in range(0, 96, 8):
            chunk = frames[a:a + 8]
            for d, pose in chunk:
                ovol.integrate(d, pose)
                assert f.integrate(d, pose)
            dev = torch.from_numpy(np.stack([d for d, _ in chunk]).view(np.int16)).cuda()
            g.integrate_batch_device(dev.data_ptr(), W * H * 2, np.stack([p.reshape(16) for _, p in chunk]).astype(np.float32))
            for d, pose in (chunk if (a // 8) % 2 else chunk[:6]):   # take most (every other time: all) of it back: blocks left with weight 0 everywhere are collected
                ovol.deintegrate(d, pose)
                assert f.deintegrate(d, pose)
                assert g.deintegrate(d, pose)
            fo = ovol.garbage_collect()
            assert f.garbage_collect() == fo and g.garbage_collect() == fo
            freed_total += fo
            for h in (f, g):
                st = h.stats()
                assert st["hash_slots_used"] == st["blocks_allocated"] == ovol.num_blocks and st["alloc_failures"] == 0
        assert freed_total > 3000 and freed_total > ovol.num_blocks / 2   # thousands of table entries came and went
        _assert_same(ovol, f)
        _assert_same(ovol, g)


def test_invalid_pose_and_empty_depth(oracle):
    from scannet_amd import fusion
    op, gp = _mk(oracle, 160, 120, voxel=0.02, num_sdf_blocks=1 << 14)
    with fusion.Fuser(gp) as f:
        bad = np.full((4, 4), -np.inf, np.float32)
        assert f.integrate(synth.plane_frame(160, 120), bad) is False
        assert f.integrate(np.zeros((120, 160), np.uint16), np.eye(4, dtype=np.float32))
        st = f.stats()
        assert st["blocks_allocated"] == 0 and st["frames_skipped"] == 1 and st["last_frame_blocks"] == 0
        with pytest.raises(ValueError):
            f.integrate(np.zeros((10, 10), np.uint16), np.eye(4, dtype=np.float32))


def test_one_mm_voxels(oracle):
    """Config 3 flavour: 1 mm voxels / 2^22 buckets on a window of the plane (HBM-capacity stress is bench territory)."""
    from scannet_amd import fusion
    op, gp = _mk(oracle, 200, 150, voxel=0.001, hash_num_buckets=1 << 22, num_sdf_blocks=1 << 18)
    ovol = oracle.Volume(op, threads=8)
    d = synth.plane_frame(200, 150, 600)
    I = np.eye(4, dtype=np.float32)
    n = ovol.integrate(d, I)
    with fusion.Fuser(gp) as f:
        assert f.integrate(d, I)
        assert f.stats()["last_frame_blocks"] == n
        _assert_same(ovol, f)


def test_one_mm_voxels_at_full_size(oracle):
    """BASELINE configs[2] at its real size: eight full 640x480 frames of the box-room walk at 1 mm voxels / 2^22 hash buckets -- every
    ray segment spans 25-35 blocks, so allocation takes the 64^3-block LDS window (k_alloc<6, false>, one frame per workgroup) and a frame
    touches ~1.2 M tiles (4.7 GB).  Block set and voxels against the oracle, through the 16-frame pass; the one-frame schedule (persistent
    pipelined kernel, next frame's allocation on the second stream) must leave the same bytes (sha256 over coordinates and voxels)."""
    import hashlib
    from scannet_amd import fusion
    op, gp = _mk(oracle, voxel=0.001, hash_num_buckets=1 << 22, num_sdf_blocks=1 << 21)
    frames = []
    for i in range(8):
        pose = synth.trajectory_pose(3 * i, 5578)
        frames.append((synth.render_room_depth(pose, noise_frame=i), pose))
    ovol = oracle.Volume(op, threads=16)
    counts = [ovol.integrate(d, pose) for d, pose in frames]
    assert min(counts) > 500000
    import torch
    dev = torch.from_numpy(np.stack([d for d, _ in frames]).view(np.int16)).cuda()
    poses = np.stack([p.reshape(16) for _, p in frames]).astype(np.float32)
    with fusion.Fuser(gp) as f:
        f.integrate_batch_device(dev.data_ptr(), 640 * 480 * 2, poses)
        f.sync()
        st = f.stats()
        assert st["alloc_failures"] == 0 and st["last_frame_blocks"] == counts[-1] and st["blocks_allocated"] == ovol.num_blocks
        assert st["total_frame_blocks"] == sum(counts)
        _assert_same(ovol, f)
        c, v = f.export_blocks()
        want = hashlib.sha256(c.tobytes() + v.tobytes()).hexdigest()
        del c, v
    ovol.close()
    for switches in ({"batch": 1}, {"batch": 1, "pipe_overlap": 0}):
        with fusion.Fuser(gp, **switches) as g:
            g.integrate_batch_device(dev.data_ptr(), 640 * 480 * 2, poses)
            g.sync()
            c, v = g.export_blocks()
            assert hashlib.sha256(c.tobytes() + v.tobytes()).hexdigest() == want, switches
            del c, v


def test_spec_literal_tolerance_on_baseline_frames():
    """The north-star tolerance, measured: the HIP volume of eight 640x480 / 4 mm frames of the configs[1] stream against the float64,
    division-form, statement-by-statement evaluation of SURVEY App. C (oracle/spec_literal.py -- independent of tsdf_oracle.c and of the
    kernel's fmaf / reciprocal forms).  Weights equal wherever no pixel cast or skip test sits within fp32 rounding distance of its
    threshold; TSDF values within 1e-4 m (measured ~5e-7)."""
    from scannet_amd import fusion
    from tests.test_oracle_tsdf import _births, spec_literal_check
    W, H = 640, 480
    fx, fy, mx, my = synth.intrinsics(W, H)
    gp = fusion.default_params(num_sdf_blocks=1 << 18)
    frames, after = [], []
    with fusion.Fuser(gp) as f:
        for i in (0, 1, 2, 3, 700, 701, 1400, 1401):
            pose = synth.trajectory_pose(i, 5578)
            d = synth.render_room_depth(pose, noise_frame=i)
            assert f.integrate(d, pose)
            frames.append((d, pose))
            after.append(f.export_blocks()[0])
        coords, vox = f.export_blocks()
    final, birth = _births(after)
    assert np.array_equal(final, coords)
    res = spec_literal_check(frames, coords, birth, vox, dict(voxel=0.004, fx=fx, fy=fy, mx=mx, my=my, width=W, height=H), sample=12000)
    assert res["max_abs_sdf_err_m"] < 1e-5, res
    print("spec-literal check:", res)


def test_integration_size_resample(oracle):
    """s_integrationWidth / s_integrationHeight (zParametersScanNet.txt:20-21, shipped as 320 x 240): 640x480 frames handed to a fuser that
    integrates at 320x240.  The pre-pass resamples (nearest, x_in = (uint)(x * (W_in - 1) / (W - 1) + 0.5f)), the intrinsics follow; the
    volume must be bit-identical to the oracle fed the same resample done in numpy -- geometry and colour (colour is looked up in the
    full-resolution image under the integration pixel's ray)."""
    from scannet_amd import fusion
    Wi, Hi, W, H = 640, 480, 320, 240
    fx, fy, mx, my = (np.float32(v) for v in synth.intrinsics(Wi, Hi))
    sx = np.float32(Wi - 1) / np.float32(W - 1)
    sy = np.float32(Hi - 1) / np.float32(H - 1)
    xi = (np.arange(W, dtype=np.float32) * sx + np.float32(0.5)).astype(np.uint32)
    yi = (np.arange(H, dtype=np.float32) * sy + np.float32(0.5)).astype(np.uint32)
    fxs, fys = fx * (np.float32(W) / np.float32(Wi)), fy * (np.float32(H) / np.float32(Hi))
    mxs, mys = mx * (np.float32(W - 1) / np.float32(Wi - 1)), my * (np.float32(H - 1) / np.float32(Hi - 1))
    op = oracle.default_params(W, H, 0.01)
    op.fx, op.fy, op.mx, op.my = float(fxs), float(fys), float(mxs), float(mys)
    gp = fusion.default_params(depth_width=Wi, depth_height=Hi, fx=float(fx), fy=float(fy), mx=float(mx), my=float(my), voxel_size=0.01,
                               integration_width=W, integration_height=H, num_sdf_blocks=1 << 17)
    rng = np.random.default_rng(3)
    for colour in (False, True):
        ovol = oracle.Volume(op, threads=8)
        with fusion.Fuser(gp) as f:
            for i in (0, 20, 40, 41):
                pose = synth.trajectory_pose(i, 400)
                d = synth.render_room_depth(pose, Wi, Hi, noise_frame=i)
                rgb = rng.integers(0, 256, (Hi, Wi, 3), dtype=np.uint8) if colour else None
                ds = d[yi][:, xi]
                rs = None
                if colour:   # the colour pixel under the integration pixel's ray, in the full-resolution image
                    # u = fmaf((x - mx') / fx', fx, mx) + 0.5f: the fused step emulated in binary64 (the product of two floats is exact there)
                    kx = (np.arange(W, dtype=np.float32) - mxs) / fxs
                    ky = (np.arange(H, dtype=np.float32) - mys) / fys
                    u = (kx.astype(np.float64) * np.float64(fx) + np.float64(mx)).astype(np.float32) + np.float32(0.5)
                    v = (ky.astype(np.float64) * np.float64(fy) + np.float64(my)).astype(np.float32) + np.float32(0.5)
                    uu, vv = np.meshgrid(u, v)
                    ok = (uu >= 0) & (uu < Wi) & (vv >= 0) & (vv < Hi)
                    rs = np.where(ok[..., None], rgb[np.clip(vv.astype(np.int64), 0, Hi - 1), np.clip(uu.astype(np.int64), 0, Wi - 1)], 0).astype(np.uint8)
                n = ovol.integrate(ds, pose, rgb=rs)
                assert f.integrate(d, pose, rgb=rgb)
                assert f.stats()["last_frame_blocks"] == n
            _assert_same(ovol, f)
        ovol.close()


def test_heap_exhaustion_is_reported(oracle):
    from scannet_amd import fusion
    _, gp = _mk(oracle, 160, 120, voxel=0.004, num_sdf_blocks=256)
    with fusion.Fuser(gp) as f:
        assert f.integrate(synth.plane_frame(160, 120, 1000), np.eye(4, dtype=np.float32))
        st = f.stats()
        assert st["heap_free"] == 0 and st["blocks_allocated"] == 256 and st["alloc_failures"] > 0


def test_full_size_roundtrip_property(oracle):
    """BASELINE size (640x480, 4 mm): integrate 8 noisy frames then deintegrate them in reverse => empty volume,
    and integrating the same stream through the device-resident batch entry point gives the same voxels."""
    import torch
    from scannet_amd import fusion
    _, gp = _mk(oracle, num_sdf_blocks=1 << 19)
    frames = [(synth.render_room_depth(synth.trajectory_pose(i, 5578), noise_frame=i), synth.trajectory_pose(i, 5578)) for i in range(0, 64, 8)]
    with fusion.Fuser(gp) as f, fusion.Fuser(gp) as g:
        for d, p in frames:
            assert f.integrate(d, p)
        dev = torch.from_numpy(np.stack([d for d, _ in frames]).astype(np.int16)).cuda()
        g.integrate_batch_device(dev, 640 * 480 * 2, np.stack([p for _, p in frames]))
        c1, v1 = f.export_blocks()
        c2, v2 = g.export_blocks()
        assert np.array_equal(c1, c2) and np.array_equal(v1.view(np.uint8), v2.view(np.uint8))
        assert f.stats()["total_frame_blocks"] == g.stats()["total_frame_blocks"]
        for d, p in reversed(frames):
            assert f.deintegrate(d, p)
        _, v = f.export_blocks()
        assert (v["w"] == 0).all() and (v["sdf"] == 0).all()
        assert f.garbage_collect() == len(c1)
        assert f.stats()["blocks_allocated"] == 0


def _assert_same_mesh(ovol, fuser):
    om = ovol.extract_mesh()
    gm = fuser.extract_mesh()
    xyz, rgba, tris, keys = gm.arrays(keys=True)
    assert len(xyz) == len(om["pos"]) and len(tris) == len(om["idx"])
    assert np.array_equal(keys, om["keys"])
    assert np.array_equal(xyz.view(np.uint32), om["pos"].view(np.uint32))  # bit-exact positions (north star asks 1e-4 m)
    assert np.array_equal(rgba[:, :3], om["col"]) and (rgba[:, 3] == 255).all()
    assert np.array_equal(tris.astype(np.int32), om["idx"])
    return gm


def test_marching_cubes_bit_exact(oracle, tmp_path):
    """Mesh extraction vs the oracle on a coloured room walk: identical vertices, colours and triangles; PLY round trip."""
    from scannet_amd import fusion, segmentator
    W, H = 320, 240
    op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 17)
    ovol = oracle.Volume(op, threads=8)
    rng = np.random.default_rng(1)
    with fusion.Fuser(gp) as f:
        # empty volume => empty mesh
        assert f.extract_mesh().counts() == (0, 0)
        for i in (0, 20, 40, 300):
            pose = synth.trajectory_pose(i, 1200)
            d = synth.render_room_depth(pose, W, H, noise_frame=i)
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            ovol.integrate(d, pose, rgb=rgb)
            assert f.integrate(d, pose, rgb=rgb)
        gm = _assert_same_mesh(ovol, f)
        nv, nf = gm.counts()
        assert nv > 20000 and nf > 40000
        p = str(tmp_path / "scene_vh.ply")
        gm.write_ply(p)
        xyz, rgba, tris = gm.arrays()
        xyz2, rgba2, tris2 = segmentator.Mesh.read(p).arrays()
        assert np.array_equal(xyz, xyz2) and np.array_equal(rgba, rgba2) and np.array_equal(tris, tris2)
        # the extracted mesh feeds the Segmentator (the pipeline's next stage): labels are well-formed roots
        seg = segmentator.segment(p, 0.01, 20)
        assert seg.shape == (nv,) and seg.min() >= 0 and seg.max() < nv and (seg[seg] == seg).all()


def test_marching_cubes_plane_full_size(oracle):
    from scannet_amd import fusion
    op, gp = _mk(oracle)
    ovol = oracle.Volume(op, threads=8)
    I = np.eye(4, dtype=np.float32)
    d = synth.plane_frame()
    ovol.integrate(d, I)
    with fusion.Fuser(gp) as f:
        assert f.integrate(d, I)
        gm = _assert_same_mesh(ovol, f)
        xyz, _, _ = gm.arrays()
        assert np.abs(xyz[:, 2] - 2.0).max() < 1e-6


def test_hand_expanded_divisions_on_device():
    """k_integrate expands its two IEEE divisions by hand (rcp + Newton; table reciprocal + Markstein correction), k_alloc its twelve per ray
    (reciprocal + two residual corrections).  On the device itself: identical bits to the hardware division for every mantissa / every
    divisor / 2^28 general operand pairs incl. near-exact and near-half-way quotients (tools/check_division.c is the CPU-side proof sketch;
    this closes the loop on the real v_rcp_f32)."""
    import ctypes as C
    from scannet_amd import _abi
    L = _abi.lib()
    a, b, c = C.c_uint64(1), C.c_uint64(1), C.c_uint64(1)
    L.sf_selftest_division.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _abi.check(L.sf_selftest_division(0, C.byref(a), C.byref(b), C.byref(c)))
    assert (a.value, b.value, c.value) == (0, 0, 0)


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_gpu_matches_committed_digests(case):
    """The HIP path against tests/golden/tsdf_golden.json (digests of the oracle's voxels and canonical mesh, generated by
    tests/golden/make_tsdf_golden.py) -- no oracle call on the GPU box for this one."""
    from scannet_amd import fusion
    from tests.test_oracle_tsdf import _golden
    mod, gold = _golden()
    name, size, voxel, idx, colour, deint, opts = mod.SCENARIOS[case]

    class Adapter:
        def __init__(self, W, H, vx, **switches):
            fx, fy, mx, my = synth.intrinsics(W, H)
            self.f = fusion.Fuser(fusion.default_params(depth_width=W, depth_height=H, voxel_size=vx, fx=fx, fy=fy, mx=mx, my=my, num_sdf_blocks=1 << 16, **switches))

        def integrate(self, d, pose, rgb=None):
            self.f.integrate(d, pose, rgb=rgb)

        def deintegrate(self, d, pose, rgb=None):
            self.f.deintegrate(d, pose, rgb=rgb)

        def export(self):
            return self.f.export_blocks()

        def extract_mesh(self):
            xyz, rgba, tris, keys = self.f.extract_mesh().arrays(keys=True)
            return xyz, np.ascontiguousarray(rgba[:, :3]), tris.astype(np.int32), keys

    got = mod.run(Adapter, name, size, voxel, idx, colour, deint, opts)
    assert got == gold[name]


def test_colour_at_its_own_resolution(oracle):
    """Real ScanNet scans store 1296x968 colour over 640x480 depth: sf_params.color_width/height + colour intrinsics make the
    pre-pass sample the colour pixel under each depth pixel's ray (nearest, black outside).  Checked against the oracle fed
    with the same resampling done in numpy."""
    from scannet_amd import fusion
    W, H, CW, CH = 160, 120, 324, 242
    op, gp = _mk(oracle, W, H, voxel=0.02, num_sdf_blocks=1 << 15)
    cfx, cfy, cmx, cmy = np.float32(340.3), np.float32(338.1), np.float32(160.2), np.float32(119.7)   # narrower than the depth camera: black rim
    gp.color_width, gp.color_height, gp.cfx, gp.cfy, gp.cmx, gp.cmy = CW, CH, cfx, cfy, cmx, cmy
    ovol = oracle.Volume(op, threads=4)
    rng = np.random.default_rng(3)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    u = (((xs - np.float32(gp.mx)) / np.float32(gp.fx)) * cfx + cmx).astype(np.float32)   # fmaf == mul + add here?  use float64 fma emulation below
    # exact fmaf: products of two floats fit in float64, one rounding to float32 at the end
    u = (((xs - np.float32(gp.mx)) / np.float32(gp.fx)).astype(np.float64) * np.float64(cfx) + np.float64(cmx)).astype(np.float32) + np.float32(0.5)
    v = (((ys - np.float32(gp.my)) / np.float32(gp.fy)).astype(np.float64) * np.float64(cfy) + np.float64(cmy)).astype(np.float32) + np.float32(0.5)
    ok = (u >= 0) & (u < CW) & (v >= 0) & (v < CH)
    iu, iv = np.where(ok, u, 0).astype(np.int64), np.where(ok, v, 0).astype(np.int64)
    assert ok.mean() < 1.0 and ok.mean() > 0.5
    with fusion.Fuser(gp) as f:
        for i in (0, 25, 50):
            pose = synth.trajectory_pose(i, 400)
            d = synth.render_room_depth(pose, W, H, noise_frame=i)
            big = rng.integers(1, 256, (CH, CW, 3), dtype=np.uint8)
            small = np.where(ok[..., None], big[iv, iu], 0).astype(np.uint8)
            ovol.integrate(d, pose, rgb=small)
            assert f.integrate(d, pose, rgb=big)
        _assert_same(ovol, f)
        with pytest.raises(ValueError):
            f.integrate(d, pose, rgb=small)   # wrong size: the fuser expects colour-resolution frames


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_one_frame_kernels_on_sparse_and_ragged_scenes(oracle, seed, monkeypatch):
    """The persistent one-frame kernel (k_integrate_pipe) walks the list with a stride of one grid (3072 waves): scenes with fewer
    tiles than waves, 1..3 tiles per wave and ragged tails exercise every branch of its hand-counted waits.  Random voxel size,
    weight step and pose; the same frames through k_integrate (tune pipe=0), with and without the second stream, and through the oracle."""
    from scannet_amd import fusion
    rng = np.random.default_rng(100 + seed)
    W, H = 160, 120
    voxel = float(rng.choice([0.004, 0.008, 0.016, 0.02]))
    ws = int(rng.choice([1, 1, 2, 5]))
    op, gp = _mk(oracle, W, H, voxel, weight_sample=ws, num_sdf_blocks=1 << 16)
    ovol = oracle.Volume(op, threads=8)
    frames = []
    for k in range(6):
        pose = synth.trajectory_pose(int(rng.integers(0, 400)), 400)
        d = synth.render_room_depth(pose, W, H, noise_frame=k).copy()
        mode = (seed + k) % 4
        if mode == 0:                       # only a small patch is valid: a handful of blocks
            keep = np.zeros_like(d, bool)
            y0, x0 = int(rng.integers(0, H - 12)), int(rng.integers(0, W - 12))
            keep[y0:y0 + int(rng.integers(1, 12)), x0:x0 + int(rng.integers(1, 12))] = True
            d[~keep] = 0
        elif mode == 1:                     # random holes
            d[rng.random(d.shape) < 0.5] = 0
        elif mode == 2:                     # a single pixel
            v = d[H // 2, W // 2]
            d[:] = 0
            d[H // 2, W // 2] = v
        frames.append((d, pose))
        ovol.integrate(d, pose)
    with fusion.Fuser(gp) as f:
        for d, pose in frames:
            f.integrate(d, pose)
        _assert_same(ovol, f)
        a_c, a_v = f.export_blocks()
    for switches in ({"pipe": 0}, {"overlap": 0}, {"pipe": 0, "overlap": 0}):
        with fusion.Fuser(gp, **switches) as g:
            for d, pose in frames:
                g.integrate(d, pose)
            b_c, b_v = g.export_blocks()
        assert np.array_equal(a_c, b_c) and np.array_equal(a_v.view(np.uint8), b_v.view(np.uint8)), switches


def test_four_schedules_one_volume_at_full_size():
    """600 frames of the 640x480 benchmark stream fused four ways -- one frame per pass through the persistent pipelined kernel (with the
    next frame's allocation on a second stream, and on one stream), one frame per pass through k_integrate, 16 frames per pass -- must leave byte-identical volumes (sha256 over coordinates and
    voxels): the size-independent property behind the bit-exactness claims at BASELINE's full size."""
    import ctypes as C
    import hashlib
    from scannet_amd import _abi, fusion
    W, H, N = 640, 480, 600
    L = _abi.lib()
    dptr = C.c_void_p()
    _abi.check(L.sf_device_malloc(0, N * W * H * 2, C.byref(dptr)))
    try:
        poses = np.zeros((N, 16), np.float32)
        _abi.check(L.sf_synth_room_device(dptr, W * H * 2, 0, N, 5578, W, H, 1, poses.ctypes.data_as(C.c_void_p)))
        params = fusion.default_params()
        digests = []
        for switches in ({"batch": 1}, {"batch": 1, "pipe": 0}, {"batch": 1, "overlap": 0}, {}):
            with fusion.Fuser(params, **switches) as f:
                f.integrate_batch_device(dptr.value, W * H * 2, poses)
                f.sync()
                c, v = f.export_blocks()
                digests.append((len(c), hashlib.sha256(c.tobytes() + v.tobytes()).hexdigest()))
        assert digests[0][0] > 50000 and len(set(digests)) == 1, digests
    finally:
        L.sf_device_free(dptr)


def test_gpu_mesh_matches_the_literal_marching_cubes():
    """The HIP mesh against the independent float64 marching cubes (oracle/spec_literal_mc.py) over the volume the HIP path itself exports:
    the same vertices (grid edges), positions within 1e-6 m, colours within one level, the same number of triangles -- no call into
    mc_oracle.c, whose agreement with the kernel would otherwise be the only evidence for the stage."""
    from scannet_amd import fusion
    from tests.test_oracle_tsdf import spec_literal_mc_check
    W, H = 320, 240
    fx, fy, mx, my = synth.intrinsics(W, H)
    gp = fusion.default_params(depth_width=W, depth_height=H, voxel_size=0.01, fx=fx, fy=fy, mx=mx, my=my, num_sdf_blocks=1 << 17)
    rng = np.random.default_rng(8)
    with fusion.Fuser(gp) as f:
        for i in range(10):
            pose = synth.trajectory_pose(40 * i, 600)
            assert f.integrate(synth.render_room_depth(pose, W, H, noise_frame=i), pose, rgb=rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
        coords, vox = f.export_blocks()
        xyz, rgba, tris, keys = f.extract_mesh().arrays(keys=True)
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    res = spec_literal_mc_check(coords[order], vox[order], dict(keys=keys, pos=xyz, col=rgba[:, :3], idx=tris), 0.01, gp.mc_thresh_factor)
    assert res["vertices"] > 20000, res
    print("literal marching cubes:", res)


# ---------------------------------------------------------------------------------------------------------------------------------------
# The regime the benchmark runs in: weights at the clamp (zParametersScanNet.txt:52-53: weight sample 1, weight max 99999999 -> 255 in the
# uchar).  After 255 observations of a voxel every further frame goes through the saturation branch of fuse_update (add-with-carry on the
# packed word for the shipped parameters, the explicit clamp otherwise) and the reciprocal table at m = 256.
# ---------------------------------------------------------------------------------------------------------------------------------------
class _DeviceFrames:
    """u16 depth frames (and optionally rgb) uploaded once; integrate_batch_device over a range of them."""

    def __init__(self, depth, rgb=None):
        import ctypes as C
        from scannet_amd import _abi
        self.L, self.C, self._abi = _abi.lib(), C, _abi
        self.depth = np.ascontiguousarray(depth, np.uint16)
        self.rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
        self.d = C.c_void_p()
        self.c = C.c_void_p()
        _abi.check(self.L.sf_device_malloc(0, self.depth.nbytes, C.byref(self.d)))
        _abi.check(self.L.sf_device_upload(self.d, self.depth.ctypes.data_as(C.c_void_p), self.depth.nbytes))
        if self.rgb is not None:
            _abi.check(self.L.sf_device_malloc(0, self.rgb.nbytes, C.byref(self.c)))
            _abi.check(self.L.sf_device_upload(self.c, self.rgb.ctypes.data_as(C.c_void_p), self.rgb.nbytes))
        self.dstride = self.depth[0].nbytes
        self.cstride = 0 if self.rgb is None else self.rgb[0].nbytes

    def fuse(self, fuser, poses, a, b):
        fuser.integrate_batch_device(self.d.value + a * self.dstride, self.dstride, poses[a:b],
                                     None if self.rgb is None else self.c.value + a * self.cstride, self.cstride)

    def close(self):
        self.L.sf_device_free(self.d)
        if self.c:
            self.L.sf_device_free(self.c)


def _static_view_stream(W, H, n, wobble_every=0, seed=11, colour=False):
    """n frames of ONE view of the box room (so every visible voxel is observed n times) with fresh sensor noise per frame; every
    `wobble_every`-th frame looks from elsewhere (new blocks are born in the middle of passes, old ones drop out of the frustum)."""
    rng = np.random.default_rng(seed)
    depth = np.zeros((n, H, W), np.uint16)
    poses = np.zeros((n, 16), np.float32)
    rgb = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8) if colour else None
    for i in range(n):
        k = 40 if not (wobble_every and i % wobble_every == wobble_every - 1) else 40 + 3 * (i // wobble_every + 1)
        pose = synth.trajectory_pose(k, 400)
        depth[i] = synth.render_room_depth(pose, W, H, noise_frame=i)
        poses[i] = pose.reshape(16)
    if colour:
        rgb[:, : H // 3] = 0   # a black band: sf_params::colour_first tells "no colour yet" from "black" differently
    return depth, poses, rgb


@pytest.mark.parametrize("schedule,colour", [("batch32", False), ("batch16", False), ("pipe", False), ("one_frame_kernel", False), ("batch32", True), ("batch16", True),
                                             ("one_frame_kernel", True)])
def test_weights_at_the_clamp_match_the_oracle(oracle, schedule, colour):
    """320 frames onto one 160x120 view: weights pass 255.  Compared bit for bit with the oracle below the clamp (130 frames), right at it (256)
    and 64 frames beyond, through the 32- and the 16-frame pass, the persistent one-frame kernel (k_integrate_pipe) and k_integrate at one frame per launch."""
    from scannet_amd import fusion
    W, H, N = 160, 120, 320
    op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 15)
    depth, poses, rgb = _static_view_stream(W, H, N, wobble_every=37, colour=colour)
    tune = {"batch32": {}, "batch16": {"batch": 16}, "pipe": {"batch": 1}, "one_frame_kernel": {"batch": 1, "pipe": 0}}[schedule]
    ovol = oracle.Volume(op, threads=8)
    dev = _DeviceFrames(depth, rgb)
    try:
        with fusion.Fuser(gp, **tune) as f:
            a = 0
            for b in (130, 256, N):
                for i in range(a, b):
                    ovol.integrate(depth[i], poses[i].reshape(4, 4), rgb=None if rgb is None else rgb[i])
                dev.fuse(f, poses, a, b)
                _assert_same(ovol, f)
                a = b
            _, gv = f.export_blocks()
            assert (gv["w"] == 255).mean() > 0.3, "the test is supposed to sit at the clamp"
            assert f.stats()["alloc_failures"] == 0
            # and back down from the clamp: 255 - 1, the mean un-weighted with the weight the voxel HOLDS (not the number of frames it saw)
            for i in (N - 1, N - 2, 200):
                ovol.deintegrate(depth[i], poses[i].reshape(4, 4), rgb=None if rgb is None else rgb[i])
                assert f.deintegrate(depth[i], poses[i].reshape(4, 4), rgb=None if rgb is None else rgb[i])
            _assert_same(ovol, f)
    finally:
        dev.close()


@pytest.mark.parametrize("weight_max", [3, 7])
@pytest.mark.parametrize("weight_sample", [1, 2, 5])
def test_small_weight_limits_match_the_oracle(oracle, weight_max, weight_sample):
    """weight_max in {3, 7} x weight_sample in {1, 2, 5}: the clamp of the generic weight path (weight + sample > max after one or two frames,
    sample > max included), 16 frames per pass and one per launch, then deintegration from the clamped weight."""
    from scannet_amd import fusion
    W, H, N = 160, 120, 20
    depth, poses, _ = _static_view_stream(W, H, N, wobble_every=7)
    for tune in ({}, {"batch": 1}, {"batch": 1, "pipe": 0}):
        op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 15, weight_max=weight_max, weight_sample=weight_sample)
        ovol = oracle.Volume(op, threads=8)
        dev = _DeviceFrames(depth)
        try:
            with fusion.Fuser(gp, **tune) as f:
                for i in range(N):
                    ovol.integrate(depth[i], poses[i].reshape(4, 4))
                dev.fuse(f, poses, 0, N)
                _assert_same(ovol, f)
                _, gv = f.export_blocks()
                assert gv["w"].max() == weight_max
                for i in (N - 1, 3):
                    ovol.deintegrate(depth[i], poses[i].reshape(4, 4))
                    assert f.deintegrate(depth[i], poses[i].reshape(4, 4))
                _assert_same(ovol, f)
        finally:
            dev.close()


def test_weights_at_the_clamp_at_full_size(oracle):
    """BASELINE configs[1] geometry (640x480, 4 mm, default parameters): 288 frames of one view through the 16-frame pass -- the kernel, the
    frame size, the voxel size and the weight regime bench.py times -- bit for bit against the oracle."""
    from scannet_amd import fusion
    W, H, N = 640, 480, 288
    op, gp = _mk(oracle, W, H, num_sdf_blocks=1 << 18)
    rng = np.random.default_rng(5)
    pose = synth.trajectory_pose(700, 5578)
    base = synth.render_room_depth(pose, W, H).astype(np.int32)
    depth = np.zeros((N, H, W), np.uint16)
    for i in range(N):
        depth[i] = np.where(base > 0, base + rng.integers(0, 8, base.shape), 0)   # 3 random LSBs per pixel and frame (SURVEY section 6 probe)
    poses = np.tile(pose.reshape(1, 16), (N, 1)).astype(np.float32)
    ovol = oracle.Volume(op, threads=16)
    dev = _DeviceFrames(depth)
    try:
        with fusion.Fuser(gp) as f:
            for i in range(N):
                ovol.integrate(depth[i], pose)
            dev.fuse(f, poses, 0, N)
            _assert_same(ovol, f)
            _, gv = f.export_blocks()
            assert (gv["w"] == 255).mean() > 0.3
    finally:
        dev.close()


# ---------------------------------------------------------------------------------------------------------------------------------------
# upstream-conformance switches (sf_params::frustum_mode / colour_round / colour_first / weight_mode): every mode in the
# kernels bit for bit against the same mode of the oracle
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("switches", [dict(frustum_mode=1), dict(colour_round=1), dict(colour_first=1), dict(weight_mode=1, weight_sample=10),
                                      dict(weight_mode=1), dict(frustum_mode=1, colour_round=1, colour_first=1, weight_mode=1, weight_sample=4),
                                      dict(weight_wrap=1, weight_max=99999999, weight_sample=40), dict(weight_wrap=1, weight_max=300, weight_sample=90, weight_mode=1)])
def test_conformance_switches_match_the_oracle(oracle, switches):
    from scannet_amd import fusion
    W, H, N = 320, 240, 40
    rng = np.random.default_rng(9)
    depth = np.zeros((N, H, W), np.uint16)
    poses = np.zeros((N, 16), np.float32)
    rgb = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    rgb[:, :, : W // 4] = 0   # black band
    for i in range(N):
        pose = synth.trajectory_pose(5 * i, 1200)
        depth[i] = synth.render_room_depth(pose, W, H, noise_frame=i)
        poses[i] = pose.reshape(16)
    for tune in ({}, {"batch": 1}):
        op, gp = _mk(oracle, W, H, voxel=0.008, num_sdf_blocks=1 << 17, **switches)
        ovol = oracle.Volume(op, threads=8)
        dev = _DeviceFrames(depth, rgb)
        try:
            with fusion.Fuser(gp, **tune) as f:
                for i in range(N):
                    ovol.integrate(depth[i], poses[i].reshape(4, 4), rgb=rgb[i])
                dev.fuse(f, poses, 0, N)
                _assert_same(ovol, f)
                for i in (N - 1, 10):
                    ovol.deintegrate(depth[i], poses[i].reshape(4, 4), rgb=rgb[i])
                    assert f.deintegrate(depth[i], poses[i].reshape(4, 4), rgb=rgb[i])
                _assert_same(ovol, f)
                assert f.stats()["alloc_failures"] == 0
        finally:
            dev.close()


def test_block_centre_frustum_geometry_only_and_mesh(oracle):
    """frustum_mode 1 without colour through the persistent one-frame kernel and the 16-frame pass, then the mesh: the block sets differ from the
    default's at the image border only, and the canonical mesh of the switched volume is the oracle's."""
    from scannet_amd import fusion
    W, H, N = 320, 240, 24
    depth = np.zeros((N, H, W), np.uint16)
    poses = np.zeros((N, 16), np.float32)
    for i in range(N):
        pose = synth.trajectory_pose(9 * i, 1200)
        depth[i] = synth.render_room_depth(pose, W, H, noise_frame=i)
        poses[i] = pose.reshape(16)
    counts = {}
    for mode in (0, 1):
        for tune in ({}, {"batch": 1}):
            op, gp = _mk(oracle, W, H, voxel=0.008, num_sdf_blocks=1 << 17, frustum_mode=mode)
            ovol = oracle.Volume(op, threads=8)
            dev = _DeviceFrames(depth)
            try:
                with fusion.Fuser(gp, **tune) as f:
                    for i in range(N):
                        ovol.integrate(depth[i], poses[i].reshape(4, 4))
                    dev.fuse(f, poses, 0, N)
                    _assert_same(ovol, f)
                    counts[mode] = f.stats()["blocks_allocated"]
                    if mode == 1 and not tune:
                        _assert_same_mesh(ovol, f)
            finally:
                dev.close()
    assert 0.9 * counts[0] < counts[1] < counts[0]


def test_reset_gives_an_empty_fuser_again(oracle):
    """sf_fuser_reset: the second scan through a reset fuser equals the same scan through a fresh one (and the oracle)."""
    from scannet_amd import fusion
    W, H = 160, 120
    op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 15)
    da, pa, _ = _static_view_stream(W, H, 20, wobble_every=5, seed=1)
    db, pb, _ = _static_view_stream(W, H, 24, wobble_every=3, seed=2)
    pb[:, 3] += 0.5   # another place
    ovol = oracle.Volume(op, threads=4)
    for i in range(len(db)):
        ovol.integrate(db[i], pb[i].reshape(4, 4))
    with fusion.Fuser(gp) as f:
        for i in range(len(da)):
            f.integrate(da[i], pa[i].reshape(4, 4))
        assert f.stats()["blocks_allocated"] > 0
        f.reset()
        st = f.stats()
        assert st["blocks_allocated"] == 0 and st["hash_slots_used"] == 0 and st["frames_integrated"] == 0 and st["high_water"] == 0
        dev = _DeviceFrames(db)
        try:
            dev.fuse(f, pb, 0, len(db))
            _assert_same(ovol, f)
        finally:
            dev.close()


@pytest.mark.parametrize("alloc_ray", [1, 0])
def test_allocation_kernels_on_a_furnished_scene(oracle, alloc_ray):
    """Both allocation kernels (occupancy bitmap in ray space / as a cube anchored at the first ray) on the furnished room, where pixel tiles
    straddle depth discontinuities, with frames that jump around inside a pass (the ray-space window is laid out from the pass's first frame: the
    later ones fall outside it and take the slow path) -- the same block set and birth frames as the oracle, 16 frames per pass and one."""
    from scannet_amd import fusion
    W, H = 320, 240
    boxes = synth.clutter_boxes()
    idx = [0, 1, 2, 3, 400, 401, 4, 5, 800, 6, 7, 8, 1100, 9, 10, 11, 12, 13, 200, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34]
    depth = np.zeros((len(idx), H, W), np.uint16)
    poses = np.zeros((len(idx), 16), np.float32)
    for k, i in enumerate(idx):
        pose = synth.trajectory_pose(i, 1200)
        depth[k] = synth.render_room_depth(pose, W, H, noise_frame=i, noise=2, boxes=boxes)
        poses[k] = pose.reshape(16)
    for tune in ({}, {"batch": 1}, {"alloc_group": 4}):
        op, gp = _mk(oracle, W, H, voxel=0.008, num_sdf_blocks=1 << 17)
        ovol = oracle.Volume(op, threads=8)
        dev = _DeviceFrames(depth)
        try:
            with fusion.Fuser(gp, alloc_ray=alloc_ray, **tune) as f:
                for k in range(len(idx)):
                    ovol.integrate(depth[k], poses[k].reshape(4, 4))
                dev.fuse(f, poses, 0, len(idx))
                assert f.stats()["alloc_failures"] == 0
                _assert_same(ovol, f)
        finally:
            dev.close()


def _probe_count(f):
    import ctypes as C
    from scannet_amd import _abi
    L = _abi.lib()
    L.sf_fuser_alloc_probe_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    n = C.c_uint64(0)
    _abi.check(L.sf_fuser_alloc_probe_count(f._h, C.byref(n)))
    return n.value


@pytest.mark.parametrize("kernel", ["ray", "cube32", "cube64"])
def test_presence_cache_changes_nothing_but_the_probes(oracle, kernel):
    """The allocation kernels' presence cache (fuser_internal.h BrickCache: "this block is in the table and older than this batch") on and off, on the
    three allocation kernels (the two cube-window kernels use it; the ray-space kernel must not care), one frame per launch and 32 per pass: the same block set, birth frames (hence voxels) as the oracle either way -- also across a
    deintegration + garbage collection that takes blocks OUT of the table (the cache must forget them: the frames fused afterwards allocate them again) --
    and, one frame per launch, a fraction of the table probes."""
    from scannet_amd import fusion
    # cube64: voxels small enough for the 64^3-block window (k_alloc<6>: a ray segment of more than 20 blocks), on a small image to keep the oracle quick
    W, H = (128, 96) if kernel == "cube64" else (320, 240)
    boxes = synth.clutter_boxes()
    idx = list(range(0, 36)) + [400, 401, 402, 37, 38, 39]
    depth = np.zeros((len(idx), H, W), np.uint16)
    poses = np.zeros((len(idx), 16), np.float32)
    for k, i in enumerate(idx):
        pose = synth.trajectory_pose(i, 1200)
        depth[k] = synth.render_room_depth(pose, W, H, noise_frame=i, noise=2, boxes=boxes)
        poses[k] = pose.reshape(16)
    voxel = 0.0015 if kernel == "cube64" else 0.008
    over = dict(num_sdf_blocks=1 << 20, hash_num_buckets=1 << 20) if kernel == "cube64" else dict(num_sdf_blocks=1 << 17)
    op, gp = _mk(oracle, W, H, voxel=voxel, **over)
    ovol = oracle.Volume(op, threads=8)
    n_first = 30
    for k in range(n_first):
        ovol.integrate(depth[k], poses[k].reshape(4, 4))
    oc0, ov0 = ovol.export()
    for k in range(20, n_first):
        ovol.deintegrate(depth[k], poses[k].reshape(4, 4))
    freed = ovol.garbage_collect()
    assert freed > 0
    for k in range(n_first, len(idx)):
        ovol.integrate(depth[k], poses[k].reshape(4, 4))
    dev = _DeviceFrames(depth)
    probes = {}
    try:
        for batch in (1, 32):
            for cache in (1, 0):
                tune = dict(batch=batch, brick_cache=cache)
                if kernel != "cube64":
                    tune["alloc_ray"] = 1 if kernel == "ray" else 0
                with fusion.Fuser(gp, **tune) as f:
                    dev.fuse(f, poses, 0, n_first)
                    gc0, gv0 = f.export_blocks()
                    assert np.array_equal(oc0, gc0) and np.array_equal(ov0.view(np.uint8), gv0.view(np.uint8)), (kernel, tune)
                    probes[(batch, cache)] = _probe_count(f)
                    for k in range(20, n_first):
                        assert f.deintegrate(depth[k], poses[k].reshape(4, 4))
                    assert f.garbage_collect() == freed
                    dev.fuse(f, poses, n_first, len(idx))
                    assert f.stats()["alloc_failures"] == 0
                    _assert_same(ovol, f)
        if kernel == "ray":   # the ray-space kernel does not use the cache (its "already queued" bitmap leaves few look-ups to save) and does not count probes
            assert set(probes.values()) == {0}, probes
        else:
            assert probes[(1, 1)] * 3 < probes[(1, 0)], probes   # one frame per launch: all but the new blocks and the previous frame's are answered by the cache
            assert probes[(32, 1)] <= probes[(32, 0)], probes
    finally:
        dev.close()


@pytest.mark.parametrize("front_prio,lowest", [(-1, 0), (0, 0), (1, 0), (-1, 1)])
def test_front_streams_change_hands_between_single_frames_and_batches(oracle, front_prio, lowest):
    """The front chain of a frame that runs beside the persistent one-frame kernel goes down a second front stream that does NOT have the front stream's high priority (made on first need), that of a pass of
    several frames down the high-priority one (fuser.hip sf_input_stream): a stream that alternates between single frames and batches -- beside the kernel forced with
    pipe_overlap 1, the tile set here is far below the size that switches it on -- hands the allocation over from one front stream to the other and back, ordered by an
    event.  Same volume as the oracle and as one stream for everything, bit for bit, with the front stream chosen automatically, always the second and always the first."""
    from scannet_amd import fusion
    W, H = 320, 240
    boxes = synth.clutter_boxes()
    n = 44
    depth = np.zeros((n, H, W), np.uint16)
    poses = np.zeros((n, 16), np.float32)
    for k in range(n):
        pose = synth.trajectory_pose(3 * k, 1200)
        depth[k] = synth.render_room_depth(pose, W, H, noise_frame=k, noise=2, boxes=boxes)
        poses[k] = pose.reshape(16)
    op, gp = _mk(oracle, W, H, voxel=0.008, num_sdf_blocks=1 << 17)
    ovol = oracle.Volume(op, threads=8)
    for k in range(n):
        ovol.integrate(depth[k], poses[k].reshape(4, 4))
    # single frames, a batch of 20 (passes of 4 + 4 + 12), single frames, a batch of 9, single frames
    cuts = [(0, 1), (1, 2), (2, 3), (3, 23), (23, 24), (24, 25), (25, 34), (34, 35), (35, 36)] + [(k, k + 1) for k in range(36, n)]
    dev = _DeviceFrames(depth)
    try:
        with fusion.Fuser(gp, front_lo_lowest=lowest, pipe_overlap=1, front_prio=front_prio) as f:
            for a, b in cuts:
                dev.fuse(f, poses, a, b)
            assert f.stats()["alloc_failures"] == 0
            _assert_same(ovol, f)
            a_c, a_v = f.export_blocks()
        with fusion.Fuser(gp, overlap=0) as g:
            for a, b in cuts:
                dev.fuse(g, poses, a, b)
            b_c, b_v = g.export_blocks()
        assert np.array_equal(a_c, b_c) and np.array_equal(a_v.view(np.uint8), b_v.view(np.uint8))
    finally:
        dev.close()


def test_ray_space_allocation_at_full_size_on_the_furnished_stream(oracle):
    """640x480, 4 mm, the furnished bench stream with hashed noise and sensor holes: 48 frames through the default schedule, bit for bit."""
    from scannet_amd import fusion
    W, H, N = 640, 480, 48
    boxes = synth.clutter_boxes()
    depth = np.zeros((N, H, W), np.uint16)
    poses = np.zeros((N, 16), np.float32)
    for k in range(N):
        i = 1000 + 3 * k
        pose = synth.trajectory_pose(i, 5578)
        depth[k] = synth.render_room_depth(pose, W, H, noise_frame=i, noise=2, boxes=boxes)
        poses[k] = pose.reshape(16)
    op, gp = _mk(oracle, W, H, num_sdf_blocks=1 << 18)
    ovol = oracle.Volume(op, threads=16)
    dev = _DeviceFrames(depth)
    try:
        with fusion.Fuser(gp) as f:
            for k in range(N):
                ovol.integrate(depth[k], poses[k].reshape(4, 4))
            dev.fuse(f, poses, 0, N)
            assert f.stats()["alloc_failures"] == 0
            _assert_same(ovol, f)
    finally:
        dev.close()


def test_weight_wrap_over_a_long_stream(oracle):
    """sf_params::weight_wrap 1 with the shipped weight limit (99999999): 300 frames of one view, the weights pass 255 and start again at 0 --
    the kernels (32 frames per pass and one per launch) against the oracle, bit for bit."""
    from scannet_amd import fusion
    W, H, N = 160, 120, 300
    depth, poses, _ = _static_view_stream(W, H, N, wobble_every=41)
    for tune in ({}, {"batch": 1}):
        op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 15, weight_wrap=1, weight_max=99999999)
        ovol = oracle.Volume(op, threads=8)
        dev = _DeviceFrames(depth)
        try:
            with fusion.Fuser(gp, **tune) as f:
                for i in range(N):
                    ovol.integrate(depth[i], poses[i].reshape(4, 4))
                dev.fuse(f, poses, 0, N)
                _assert_same(ovol, f)
                _, gv = f.export_blocks()
                assert 30 <= gv["w"].max() <= 255 and (gv["w"] == (N - 256)).mean() > 0.2   # wrapped: a voxel seen by every frame holds N - 256
        finally:
            dev.close()


def test_host_entry_point_has_read_the_buffers_when_it_returns(oracle):
    """sf_fuser_integrate takes pageable buffers that are the caller's again the moment it returns (a live stream decodes the next frame into the
    same memory).  One depth and one colour buffer are overwritten right after every call -- with garbage, then with the next frame --, 40 calls in
    a row without a sync: the page-locked ring (3 slots) must have taken its copy each time, and wraps around a dozen times."""
    from scannet_amd import fusion
    W, H, N = 160, 120, 40
    op, gp = _mk(oracle, W, H, voxel=0.01, num_sdf_blocks=1 << 15)
    depth, poses, rgb = _static_view_stream(W, H, N, wobble_every=6, colour=True)
    ovol = oracle.Volume(op, threads=8)
    dbuf = np.zeros((H, W), np.uint16)
    cbuf = np.zeros((H, W, 3), np.uint8)
    with fusion.Fuser(gp) as f:
        for i in range(N):
            use_colour = i % 3 != 0        # the ring serves colour and geometry-only frames alike
            dbuf[:] = depth[i]
            cbuf[:] = rgb[i]
            ovol.integrate(depth[i], poses[i].reshape(4, 4), rgb=rgb[i] if use_colour else None)
            assert f.integrate(dbuf, poses[i].reshape(4, 4), rgb=cbuf if use_colour else None)
            dbuf[:] = 0xFFFF               # the caller scribbles over its buffers at once
            cbuf[:] = 0x55
        _assert_same(ovol, f)
        # deintegrate through the same entry point
        for i in (N - 1, N - 2):
            dbuf[:] = depth[i]
            cbuf[:] = rgb[i]
            c = i % 3 != 0
            ovol.deintegrate(depth[i], poses[i].reshape(4, 4), rgb=rgb[i] if c else None)
            assert f.deintegrate(dbuf, poses[i].reshape(4, 4), rgb=cbuf if c else None)
            dbuf[:] = 1
        _assert_same(ovol, f)


# ---------------------------------------------------------------------------------------------------------------------------------------
# RGB-D at the benchmark's size: the metric is "RGB-D frames/s at 640x480 / 4 mm" -- the colour kernel
# k_integrate<1, true, true, 2, *> at exactly that geometry, with colour at depth resolution and at ScanNet's real 1296x968 with its own
# intrinsics (sensorData.h:600-616; the vertex colours meshlabserver keeps with -m vc, scan_processor.py:143), through every schedule
# ---------------------------------------------------------------------------------------------------------------------------------------
def _baseline_walk_on_device(n, first, W=640, H=480):
    """n frames of the configs[1] walk (furnished room, hashed noise: bench.py's default input) rendered by the device renderer; returns
    (device pointer, host copy [n, H, W] u16, poses [n, 16])."""
    import ctypes as C
    from scannet_amd import _abi
    L = _abi.lib()
    dptr = C.c_void_p()
    nbytes = n * W * H * 2
    _abi.check(L.sf_device_malloc(0, nbytes, C.byref(dptr)))
    poses = synth.render_scan_device(dptr.value, W * H * 2, first, n, 5578, W, H, noise=2, scene=1, seed=0)
    host = np.zeros((n, H, W), np.uint16)
    _abi.check(L.sf_device_download(host.ctypes.data_as(C.c_void_p), dptr, nbytes))
    return dptr, host, poses


def _texture_frames(n, H, W, seed):
    """A colour frame per depth frame: per-pixel hashed bytes over a gradient that moves with the frame, a black band (the colour_first rule
    tells black from unobserved) and saturated patches (255 + 255 must not carry into the neighbouring channel)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([xx * 255 // W, yy * 255 // H, xx + yy], -1).astype(np.uint8)   # uint8 arithmetic below wraps modulo 256
    out = np.zeros((n, H, W, 3), np.uint8)
    for i in range(n):
        out[i] = base + (np.array([5 * i, 3 * i, 7 * i]) % 256).astype(np.uint8) + rng.integers(0, 32, (H, W, 3), dtype=np.uint8)
        out[i, : H // 8] = 0
        out[i, H // 2 : H // 2 + H // 16, : W // 3] = 255
    return out


@pytest.mark.parametrize("colour_size", ["depth", "1296x968"])
def test_rgbd_baseline_walk_matches_the_oracle(oracle, colour_size):
    """64 frames of the configs[1] walk round a corner of the furnished room, 640x480 depth / 4 mm voxels / shipped parameters, a colour frame per
    depth frame -- at depth resolution and at 1296x968 with the colour camera's own intrinsics -- through the 32-frame pass (first pass 8: the
    ramp), the 16-frame pass and one frame per launch: block set, sdf, rgb and weight bit for bit against oracle.Volume."""
    import ctypes as C
    from scannet_amd import _abi, fusion
    W, H, N = 640, 480, 64
    op, gp = _mk(oracle, W, H, num_sdf_blocks=1 << 18)
    dptr, depth, poses = _baseline_walk_on_device(N, 1368)
    L = _abi.lib()
    cptr = C.c_void_p()
    try:
        if colour_size == "depth":
            rgb = _texture_frames(N, H, W, seed=21)
            small = rgb
        else:
            CW, CH = 1296, 968
            cfx, cfy, cmx, cmy = np.float32(1170.19), np.float32(1170.19), np.float32(647.75), np.float32(483.75)   # a ScanNet colour camera
            gp.color_width, gp.color_height, gp.cfx, gp.cfy, gp.cmx, gp.cmy = CW, CH, cfx, cfy, cmx, cmy
            rgb = _texture_frames(N, CH, CW, seed=22)
            # the pre-pass's look-up restated in numpy: colour pixel under the depth pixel's ray, nearest, black outside (fmaf = one rounding)
            xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
            u = (((xs - np.float32(gp.mx)) / np.float32(gp.fx)).astype(np.float64) * np.float64(cfx) + np.float64(cmx)).astype(np.float32) + np.float32(0.5)
            v = (((ys - np.float32(gp.my)) / np.float32(gp.fy)).astype(np.float64) * np.float64(cfy) + np.float64(cmy)).astype(np.float32) + np.float32(0.5)
            ok = (u >= 0) & (u < CW) & (v >= 0) & (v < CH)
            iu, iv = np.where(ok, u, 0).astype(np.int64), np.where(ok, v, 0).astype(np.int64)
            assert 0.5 < ok.mean() <= 1.0
            small = np.where(ok[None, ..., None], rgb[:, iv, iu], 0).astype(np.uint8)
        ovol = oracle.Volume(op, threads=16)
        last = 0
        for i in range(N):
            last = ovol.integrate(depth[i], poses[i].reshape(4, 4), rgb=small[i])
        _abi.check(L.sf_device_malloc(0, rgb.nbytes, C.byref(cptr)))
        _abi.check(L.sf_device_upload(cptr, rgb.ctypes.data_as(C.c_void_p), rgb.nbytes))
        for tune in ({}, {"batch": 16}, {"batch": 1}):
            with fusion.Fuser(gp, **tune) as f:
                f.integrate_batch_device(dptr.value, W * H * 2, poses, cptr.value, rgb[0].nbytes)
                st = f.stats()
                assert st["alloc_failures"] == 0 and st["frames_integrated"] == N and st["last_frame_blocks"] == last
                _assert_same(ovol, f)
                _, gv = f.export_blocks()
                seen = gv["w"] > 0
                assert seen.mean() > 0.2 and (gv["r"][seen] > 0).mean() > 0.5, "colour did not reach the voxels"
        ovol.close()
    finally:
        L.sf_device_free(dptr)
        if cptr:
            L.sf_device_free(cptr)


def test_bench_walk_corner_stays_on_the_allocation_fast_path():
    """when k_alloc_ray re-anchors its window inside a group (a fast-turning tile) the blocks already queued are queued again; the
    duplicates must not fill the workgroup's LDS queue, or its blocks go to the global table one probe at a time -- the slow path the kernel was
    written to avoid.  96 frames round a corner of the bench walk (the fastest turn of the stream) at full size: the direct path is never taken."""
    import ctypes as C
    from scannet_amd import _abi, fusion
    W, H, N = 640, 480, 96
    gp = fusion.default_params(depth_width=W, depth_height=H, num_sdf_blocks=1 << 18)
    dptr, _, poses = _baseline_walk_on_device(N, 1340)
    L = _abi.lib()
    L.sf_fuser_alloc_direct_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    try:
        for tune in ({}, {"batch": 16}, {"alloc_group": 32}):
            with fusion.Fuser(gp, **tune) as f:
                f.integrate_batch_device(dptr.value, W * H * 2, poses)
                n = C.c_uint64(123)
                _abi.check(L.sf_fuser_alloc_direct_count(f._h, C.byref(n)))
                assert n.value == 0, "%d blocks took the one-by-one path under %r" % (n.value, tune)
                assert f.stats()["alloc_failures"] == 0
    finally:
        L.sf_device_free(dptr)


def test_conformance_packet_digests():
    """conformance/digests.json (tools/conformance_packet.py: the oracle's volume and mesh digests on the packet's 300-frame scan under all 32
    combinations of the upstream-conformance switches): the HIP path reproduces every entry through the C ABI -- after the 40-frame walk and after
    the dwell that takes the 8-bit weights through 255 (saturation or wrap)."""
    import importlib.util
    import json
    import os
    from scannet_amd import fusion
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("conformance_packet", os.path.join(root, "tools", "conformance_packet.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    want = json.load(open(os.path.join(root, "conformance", "digests.json")))["combinations"]
    fr = cp.frames("full")
    for sw in cp.combos():
        got = cp.run_gpu(fusion, fr, sw)
        assert got == want[cp.name_of(sw)], cp.name_of(sw)
