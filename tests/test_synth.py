"""The synthetic input generator (scannet_amd/synth.py on the host, csrc/synth.hip on the device): test inputs, not product -- but bench.py's
numbers are only as honest as its inputs (the round-1 "noise" was a ramp that deflated to 54 KB per frame where real depth
takes 100-300 KB, SensReader/c++/README.txt:32; every scene was an empty box)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from scannet_amd import synth


def test_host_and_library_draw_the_same_furniture():
    from scannet_amd import _abi
    L = _abi.lib()
    L.sf_synth_clutter_boxes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    for seed, room in ((0, synth.ROOM), (7, (4.8, 3.2, 2.4)), (1512, (7.2, 4.8, 3.6))):
        lo, hi = synth.clutter_boxes(room, seed)
        r = np.asarray(room, np.float64)
        glo, ghi = np.zeros((48, 3), np.float32), np.zeros((48, 3), np.float32)
        n = C.c_int(0)
        _abi.check(L.sf_synth_clutter_boxes(r.ctypes.data, seed, glo.ctypes.data, ghi.ctypes.data, C.byref(n)))
        assert n.value == 48 == len(lo)
        assert np.array_equal(lo, glo) and np.array_equal(hi, ghi)
        # inside the room, non-degenerate, clear of the walk (1 m inset from the walls at 1.5 m height): a box either stays within 0.65 m of a
        # wall, or inside the island, or above 1.9 m
        assert (lo >= 0).all() and (hi <= np.asarray(room, np.float32) + 1e-6).all() and (hi - lo > 0.04).all()
        for a, b in zip(lo, hi):
            near_wall = b[0] <= 0.65 or b[1] <= 0.65 or a[0] >= room[0] - 0.65 or a[1] >= room[1] - 0.65
            island = a[0] >= 1.39 and a[1] >= 1.39 and b[0] <= room[0] - 1.39 and b[1] <= room[1] - 1.39
            assert near_wall or island or a[2] >= 1.9, (a, b)


def test_hashed_noise_has_the_entropy_of_real_low_bits():
    """Per-pixel hashed noise deflates like real sensor depth (>= 115 KB per 640x480 frame at zlib level 6 -- the reference's stb deflater
    does a little worse); the round-1 ramp does not (~ 50 KB)."""
    pose = synth.trajectory_pose(700, 5578)
    boxes = synth.clutter_boxes()
    ramp = synth.render_room_depth(pose, noise_frame=700, noise=1)
    real = synth.render_room_depth(pose, noise_frame=700, noise=2, boxes=boxes)
    n_ramp, n_real = len(zlib.compress(ramp.tobytes(), 6)), len(zlib.compress(real.tobytes(), 6))
    assert n_ramp < 90_000 < 115_000 < n_real, (n_ramp, n_real)
    # the three noise bits of neighbouring pixels are independent: each value about 1/8 of the time, lag-1 agreement about 1/8
    clean = synth.render_room_depth(pose, boxes=boxes).astype(np.int64)
    ok = (real > 0) & (clean > 0)
    bits = (real.astype(np.int64) - clean)[ok]
    assert bits.min() == 0 and bits.max() == 7
    assert np.abs(np.bincount(bits, minlength=8) / bits.size - 0.125).max() < 0.01
    row = (real.astype(np.int64) - clean)
    same = (row[:, 1:] == row[:, :-1])[ok[:, 1:] & ok[:, :-1]].mean()
    assert abs(same - 0.125) < 0.02, same


def test_furnished_scene_has_clutter_holes_and_a_clear_walk():
    boxes = synth.clutter_boxes()
    empty_hits, holes, nearest = [], [], []
    for i in range(0, 5578, 279):
        pose = synth.trajectory_pose(i, 5578)
        a = synth.render_room_depth(pose, 320, 240)
        b = synth.render_room_depth(pose, 320, 240, boxes=boxes, noise_frame=i, noise=2)
        c = synth.render_room_depth(pose, 320, 240, boxes=boxes)
        empty_hits.append((c != a).mean())
        holes.append((b == 0).mean())
        nearest.append(c[c > 0].min())
    assert np.mean(empty_hits) > 0.2, "furniture should cover a good part of what the camera sees"
    assert 0.004 < np.mean(holes) < 0.2, np.mean(holes)          # speckle + grazing surfaces, not half the image
    assert min(nearest) > 300, "the walk must stay clear of the furniture (mm)"


@pytest.mark.gpu
def test_device_renderer_draws_the_host_scene():
    """csrc/synth.hip against scannet_amd/synth.py, frame by frame: the same scene (a float64 ray cast in a different summation order: identical
    but for pixels whose ray grazes an edge or whose depth sits within rounding distance of a millimetre boundary)."""
    from scannet_amd import _abi
    L = _abi.lib()
    W, H, N = 320, 240, 6
    dptr = C.c_void_p()
    _abi.check(L.sf_device_malloc(0, N * W * H * 2, C.byref(dptr)))
    try:
        for scene, noise in ((1, 2), (0, 2), (1, 0), (0, 1)):
            poses = synth.render_scan_device(dptr.value, W * H * 2, 1000, N, 5578, W, H, noise=noise, scene=scene, seed=3)
            got = np.zeros((N, H, W), np.uint16)
            _abi.check(L.sf_device_download(got.ctypes.data_as(C.c_void_p), dptr, got.nbytes))
            boxes = synth.clutter_boxes(synth.ROOM, 3) if scene == 1 else None
            for k in range(N):
                pose = synth.trajectory_pose(1000 + k, 5578)
                assert np.allclose(poses[k].reshape(4, 4), pose, atol=1e-6)
                want = synth.render_room_depth(poses[k].reshape(4, 4), W, H, boxes=boxes, noise_frame=(1000 + k) if noise else None, noise=noise or 1)
                diff = np.abs(got[k].astype(np.int64) - want.astype(np.int64))
                assert (diff == 0).mean() > 0.995 and (diff <= 1).mean() > 0.997, (scene, noise, k, (diff == 0).mean(), (diff <= 1).mean())
    finally:
        L.sf_device_free(dptr)
