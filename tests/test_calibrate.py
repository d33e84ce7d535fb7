"""Calibrate stage: per-frame image operations of Calibrate/src/calibration.h:253-307.
CPU part: the checker oracle/calib_oracle.c against closed-form cases, its own point-splat variant (the reference's
depthToColorDebug) and golden digests.  GPU part (-m gpu): scannet_amd/csrc/calibrate.hip against the checker, bit for bit.
PARITY UNPINNED against the reference binary (mLib + Direct3D 11, not buildable here; its warp is a hardware draw call)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from scannet_amd import calibrate, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, CW, CH = 640, 480, 1296, 968
K_COLOR = (1170.19, 1170.19, 647.75, 483.75)
K_DEPTH = (571.62, 571.62, 319.5, 239.5)
D2C = [[0.99998, 0.006, -0.002, -0.037], [-0.006, 0.99997, 0.004, 0.003], [0.002, -0.004, 0.99999, -0.021], [0, 0, 0, 1]]
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "calibrate_golden.json")))


def _scene(frame=40, seed=1, holes=True):
    """Depth of the synthetic room with sensor noise and a few holes; a colour image with a black band (no colour there)."""
    d = synth.render_room_depth(synth.trajectory_pose(frame, 400), W, H, noise_frame=frame).copy()
    rng = np.random.default_rng(seed)
    if holes:
        d[200:230, 300:360] = 0
        d[rng.random((H, W)) < 0.002] = 0
    rgb = rng.integers(1, 256, (CH, CW, 3), dtype=np.uint8)
    rgb[:, 900:930] = 0
    return d, rgb


def _full_params():
    return calibrate.make_params((CW, CH), (W, H), K_COLOR, K_DEPTH, D2C, color_dist=(0.05, -0.1, 0.001, -0.0005, 0.02),
                                 depth_dist=(-0.12, 0.08, 0.0008, 0.0006, -0.01))


def _lut(seed=2):
    rng = np.random.default_rng(seed)
    return (1.0 + 0.02 * rng.standard_normal((10, 48, 64))).astype(np.float32), 10.0


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------------- CPU: the checker
def test_undistort_known_answers():
    L = orc.calib_lib()
    rng = np.random.default_rng(0)
    img = rng.random((H, W), dtype=np.float32)
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = K_DEPTH
    zero = np.zeros(5, np.float32)
    out = np.empty_like(img)
    L.or_calib_undistort_f32(img.ctypes.data, out.ctypes.data, W, H, K.ctypes.data, zero.ctypes.data, 0.0)
    assert np.array_equal(out, img)                      # no distortion: identity (calibration.h:192-217 with all k = 0)
    # pure radial k1: the sample location of a pixel follows the closed form, rounded half up
    k = np.array([-0.2, 0, 0, 0, 0], np.float32)
    idx = np.arange(W * H, dtype=np.float32).reshape(H, W)   # exact in fp32 (< 2^24)
    L.or_calib_undistort_f32(idx.ctypes.data, out.ctypes.data, W, H, K.ctypes.data, k.ctypes.data, -1.0)
    for (x, y) in ((10, 20), (320, 240), (600, 400), (639, 0)):
        nx, ny = np.float32((x - K_DEPTH[2]) / K_DEPTH[0]), np.float32((y - K_DEPTH[3]) / K_DEPTH[1])
        r2 = nx * nx + ny * ny
        sx = int(np.floor(float(nx * (1 + r2 * k[0])) * K_DEPTH[0] + K_DEPTH[2] + 0.5))
        sy = int(np.floor(float(ny * (1 + r2 * k[0])) * K_DEPTH[1] + K_DEPTH[3] + 0.5))
        want = sy * W + sx if (0 <= sx < W and 0 <= sy < H) else -1
        assert out[y, x] == want
    # colour: invalid = black
    rgb = rng.integers(1, 256, (50, 60, 3), dtype=np.uint8)
    Kc = np.eye(4, dtype=np.float32)
    Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2] = 50, 50, 29.5, 24.5
    big = np.array([0.9, 0, 0, 0, 0], np.float32)          # strong barrel: the corners sample outside
    o = np.empty_like(rgb)
    L.or_calib_undistort_rgb(rgb.ctypes.data, o.ctypes.data, 60, 50, Kc.ctypes.data, big.ctypes.data)
    assert (o[0, 0] == 0).all() and np.array_equal(o[25, 30], rgb[25, 30])


def test_distance_table():
    L = orc.calib_lib()
    d = np.full((H, W), 2000, np.uint16)
    d[0, 0] = 0
    for value, want in ((1.0, 2000), (0.8, 2500), (1.25, 1600)):     # multiplier = 1 / table value (calibration.h:243)
        g = np.full((10, 48, 64), value, np.float32)
        lut = orc.OrLut(64, 48, 10, 10.0, g.ctypes.data)
        x = d.copy()
        L.or_calib_undistort_distance(x.ctypes.data, W, H, C.byref(lut), 1000.0)
        # (truncating u16 conversion, calibration.h:245: the trilinear weights sum to 1 only up to rounding, hence +-1)
        assert x[0, 0] == 0 and np.abs(x[1:, 1:].astype(int) - want).max() <= 1
    # trilinear in z: table = 1 + 0.1 z_slice; depth 2.5 m with zbin = 1 slice per metre -> 1.25
    g = (1.0 + 0.1 * np.arange(10, dtype=np.float32))[:, None, None] * np.ones((10, 48, 64), np.float32)
    lut = orc.OrLut(64, 48, 10, 10.0, g.ctypes.data)
    x = np.full((H, W), 2500, np.uint16)
    L.or_calib_undistort_distance(x.ctypes.data, W, H, C.byref(lut), 1000.0)
    assert abs(int(x[7, 9]) - 2000) <= 1


def test_warp_against_the_reference_point_splat():
    """The rasterised warp (the reference's GPU path) and depthToColorDebug (its CPU variant) must agree wherever both drew."""
    L = orc.calib_lib()
    d16, _ = _scene(holes=False)
    d = (d16.astype(np.float32) / np.float32(1000.0))
    for params in (calibrate.make_params((W, H), (W, H), K_DEPTH, K_DEPTH), calibrate.make_params((CW, CH), (W, H), K_COLOR, K_DEPTH, D2C)):
        cb = orc.calib_from(params)
        ras, spl = np.empty_like(d), np.empty_like(d)
        L.or_calib_depth_to_color(d.ctypes.data, ras.ctypes.data, W, H, C.byref(cb), 0.0)
        L.or_calib_depth_to_color_splat(d.ctypes.data, spl.ctypes.data, W, H, C.byref(cb), 0.0)
        both = (ras > 0) & (spl > 0)
        assert both.mean() > 0.9
        err = np.abs(ras - spl)[both]
        assert np.median(err) < 0.004 and np.percentile(err, 99) < 0.05    # same surface, sub-pixel different sampling
        assert (ras > 0).mean() >= (spl > 0).mean() - 0.01                  # the mesh has no pin holes where the splat has
    # identity calibration: the warp returns the input up to its half-pixel resampling
    cb = orc.calib_from(calibrate.make_params((W, H), (W, H), K_DEPTH, K_DEPTH))
    plane = np.full((H, W), 2.0, np.float32)
    out = np.empty_like(plane)
    L.or_calib_depth_to_color(plane.ctypes.data, out.ctypes.data, W, H, C.byref(cb), 0.0)
    inner = out[2:-2, 2:-2]
    assert (inner > 0).all() and np.abs(inner - 2.0).max() < 2e-6
    # a depth discontinuity is not bridged (aligner.hlsl:150-154): no value strictly between the two surfaces
    step = np.full((H, W), 1.0, np.float32)
    step[:, 320:] = 3.0
    L.or_calib_depth_to_color(step.ctypes.data, out.ctypes.data, W, H, C.byref(cb), 0.0)
    vals = out[out > 0]
    assert ((np.abs(vals - 1.0) < 1e-5) | (np.abs(vals - 3.0) < 1e-5)).all() and (out[:, 318:324] == 0).any()


def test_frame_golden():
    """The whole frame body on seeded inputs: digests pin the checker across machines (and are what the GPU must reproduce)."""
    d, rgb = _scene()
    grid, maxd = _lut()
    cb = orc.calib_from(_full_params())
    do, ro = orc.calib_frame(cb, d, rgb, grid, maxd)
    assert (do > 0).mean() > 0.9
    assert _sha(do, ro) == GOLDEN["frame_full"]
    do2, _ = orc.calib_frame(cb, d, None, None)
    assert _sha(do2) == GOLDEN["frame_depth_only"]
    # "invalidate depth where we have no color": the black band of the colour image removes a band of depth
    band = do[:, int(905 * (W - 1) / (CW - 1)) + 2:int(925 * (W - 1) / (CW - 1)) - 2]
    assert (band == 0).all() and (do2[:, 450:455] > 0).any()


def test_parameter_and_table_files(tmp_path):
    p = _full_params()
    calibrate.write_params(str(tmp_path / "dev.txt"), p)
    q = calibrate.load_params(str(tmp_path / "dev.txt"))
    for name, _ in calibrate.SfCalibParams._fields_:
        a, b = getattr(p, name), getattr(q, name)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), name
    grid, maxd = _lut()
    calibrate.write_lut(str(tmp_path / "dev.lut"), grid, maxd)
    t = calibrate.SfLut()
    L = calibrate._lib()
    assert L.sf_lut_load(str(tmp_path / "dev.lut").encode(), C.byref(t)) == 0
    assert (t.xres, t.yres, t.zres, t.max_dist) == (64, 48, 10, 10.0)
    assert np.array_equal(np.ctypeslib.as_array(t.data, (10, 48, 64)), grid)
    L.sf_lut_free(C.byref(t))
    open(tmp_path / "bad.lut", "wb").write(b"\x01\x00\x00\x00")
    assert L.sf_lut_load(str(tmp_path / "bad.lut").encode(), C.byref(t)) != 0


def test_jpeg_encoder_round_trip():
    """The colour re-encode of the stage (sensorData.h:565-596 TYPE_JPEG): a baseline JPEG every reader decodes."""
    import io
    from PIL import Image
    from scannet_amd import sens
    y, x = np.mgrid[0:97, 0:131]                        # odd sizes: partial MCUs
    img = np.stack([x * 255 // 131, y * 255 // 97, (128 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0))], -1).astype(np.uint8)
    img[20:40, 50:80] = (255, 0, 0)
    for sub, floor_db in ((True, 33.0), (False, 40.0)):
        blob = calibrate.jpeg_encode(img, 90, sub)
        assert blob[:2] == b"\xff\xd8" and blob[-2:] == b"\xff\xd9"
        pil = np.array(Image.open(io.BytesIO(blob)).convert("RGB")).astype(float)
        psnr = 10 * np.log10(255.0 ** 2 / ((pil - img) ** 2).mean())
        assert psnr > floor_db, psnr
        # ... including this repository's own decoder, through a .sens
        K = np.eye(4, dtype=np.float32)
        sd = sens.SensorData.create(131, 97, 8, 8, K, K, color_compression=2, depth_compression=1)
        sd.add_frame(np.zeros((8, 8), np.uint16), np.eye(4, dtype=np.float32), color=blob)
        mine = sd.frames[0].decompress_color().astype(float)
        assert np.abs(mine - pil).max() <= 6 and np.abs(mine - pil).mean() < 1.0
        sd.close()


# ---------------------------------------------------------------------------------------------------- GPU: parity
@pytest.mark.gpu
def test_gpu_matches_the_checker_bit_for_bit():
    grid, maxd = _lut()
    p = _full_params()
    cb = orc.calib_from(p)
    frames = [_scene()] + [_scene(frame=f, seed=f) for f in (41, 200)]   # frame 0 = the scene of the golden digests
    depth = np.stack([f[0] for f in frames])
    rgb = np.stack([f[1] for f in frames])
    with calibrate.Calibrator(p, grid, maxd) as cal:
        dout, rout = cal.run(depth, rgb)
        for i in range(len(frames)):
            do, ro = orc.calib_frame(cb, depth[i], rgb[i], grid, maxd)
            assert np.array_equal(rout[i], ro), "colour frame %d" % i
            assert np.array_equal(dout[i], do), "depth frame %d: %d pixels differ" % (i, (dout[i] != do).sum())
        assert _sha(dout[0], rout[0]) == GOLDEN["frame_full"]
        d_only, none = cal.run(depth[:1])               # no colour: nothing invalidated
        assert none is None
    with calibrate.Calibrator(p) as cal:                # no table
        d_only, _ = cal.run(depth[:1])
        assert np.array_equal(d_only[0], orc.calib_frame(cb, depth[0])[0]) and _sha(d_only[0]) == GOLDEN["frame_depth_only"]


@pytest.mark.gpu
def test_gpu_edge_cases():
    # identity calibration at equal resolutions, a full batch of 16, an all-invalid frame, a small odd-sized image
    p = calibrate.make_params((W, H), (W, H), K_DEPTH, K_DEPTH)
    cb = orc.calib_from(p)
    frames = np.stack([_scene(frame=10 * i, seed=i)[0] for i in range(16)])
    frames[5] = 0
    with calibrate.Calibrator(p) as cal:
        out, _ = cal.run(frames)
        for i in (0, 5, 15):
            assert np.array_equal(out[i], orc.calib_frame(cb, frames[i])[0])
        assert (out[5] == 0).all()
        with pytest.raises(Exception):
            cal.run(np.zeros((17, H, W), np.uint16))
    w, h, cw, ch = 37, 29, 53, 41
    p = calibrate.make_params((cw, ch), (w, h), (60.0, 61.0, 26.0, 20.0), (40.0, 41.0, 18.0, 14.0), color_dist=(0.1, 0, 0, 0, 0), depth_dist=(-0.05, 0, 0.001, 0, 0))
    cb = orc.calib_from(p)
    rng = np.random.default_rng(9)
    d = (1500 + 40 * np.add.outer(np.arange(h), np.arange(w))).astype(np.uint16)
    rgb = rng.integers(0, 256, (ch, cw, 3), dtype=np.uint8)
    with calibrate.Calibrator(p) as cal:
        dout, rout = cal.run(d[None], rgb[None])
        do, ro = orc.calib_frame(cb, d, rgb)
        assert np.array_equal(rout[0], ro) and np.array_equal(dout[0], do)


@pytest.mark.gpu
def test_stage_and_cli(tmp_path):
    """calibrate.exe in.sens out.sens devices.csv devices_dir (Server/scan_processor.py:118): header rewrite, frames, file handling."""
    import subprocess
    from scannet_amd import sens
    p = _full_params()
    cb = orc.calib_from(p)
    grid, maxd = _lut()
    scan = tmp_path / "scan7"
    scan.mkdir()
    devdir = tmp_path / "devices"
    devdir.mkdir()
    calibrate.write_params(str(devdir / "structure_A.txt"), p)
    calibrate.write_lut(str(devdir / "structure_A.lut"), grid, maxd)
    (tmp_path / "devices.csv").write_text("name,id,calibration_name\nfoo,dev-123,structure_A\nbar,dev-999,other\n")
    (scan / "scan7.txt").write_text("colorWidth = 1296\r\ndeviceId = dev-123\r\nnumDepthFrames = 3\r\n")
    Kc = np.array(p.color_intrinsic, np.float32).reshape(4, 4)
    Kd = np.array(p.depth_intrinsic, np.float32).reshape(4, 4)
    n = 3
    frames = []
    yy, xx = np.mgrid[0:CH, 0:CW]
    for i in range(n):                                           # a compressible colour image (per-pixel noise is not JPEG's domain)
        d, _ = _scene(frame=40 + i, seed=i + 1)
        rgb = np.stack([60 + xx * 150 // CW + 10 * i, 40 + yy * 180 // CH, 128 + 90 * np.sin(xx / 40.0) * np.cos(yy / 55.0)], -1).astype(np.uint8)
        rgb[:, 900:930] = 0
        frames.append((d, rgb))
    sd = sens.SensorData.create(CW, CH, W, H, Kc, Kd, color_compression=2, depth_compression=1, sensor_name="StructureSensor",
                                extrinsic_depth=np.array(p.depth_extrinsic, np.float32).reshape(4, 4))
    poses = [np.eye(4, dtype=np.float32) for _ in range(n)]
    poses[1][0, 3] = 0.25
    for i, (d, rgb) in enumerate(frames):
        sd.add_frame(d, poses[i], color=calibrate.jpeg_encode(rgb, 95, False), timestamp_color=1000 + i, timestamp_depth=2000 + i)
    src = str(scan / "scan7.uncalibrated.sens")
    sd.save(src)
    decoded = [fr.decompress_color() for fr in sd.frames]     # what the stage sees after JPEG decode
    sd.close()
    dst = str(scan / "scan7.sens")
    exe = os.path.join(ROOT, "bin", "calibrate")
    r = subprocess.run([exe, src, dst, str(tmp_path / "devices.csv"), str(devdir)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    assert "calibration name: structure_A" in r.stdout and "3 frames" in r.stdout
    assert os.path.exists(dst) and not os.path.exists(src)                  # calibration.h:135: the input is consumed
    out = sens.SensorData(dst)
    assert out.sensor_name == "StructureSensor (calibrated)" and out.num_frames == n
    assert np.array_equal(out.extrinsic_depth, np.eye(4, dtype=np.float32)) and np.array_equal(out.extrinsic_color, np.eye(4, dtype=np.float32))
    assert np.array_equal(out.intrinsic_color, Kc)
    want = Kc.copy()                                                         # calibration.h:123-128
    want[0, 0] *= np.float32(W) / np.float32(CW)
    want[1, 1] *= np.float32(H) / np.float32(CH)
    want[0, 2] *= np.float32(W - 1) / np.float32(CW - 1)
    want[1, 2] *= np.float32(H - 1) / np.float32(CH - 1)
    assert np.array_equal(out.intrinsic_depth, want)
    for i, fr in enumerate(out.frames):
        do, ro = orc.calib_frame(cb, frames[i][0], decoded[i], grid, maxd)
        assert np.array_equal(fr.decompress_depth(), do), i
        got = fr.decompress_color().astype(float)                            # re-encoded at quality 90, 4:2:0: close to the undistorted image
        assert 10 * np.log10(255.0 ** 2 / ((got - ro) ** 2).mean()) > 30.0
        assert np.array_equal(fr.camera_to_world, poses[i]) and (fr.timestamp_color, fr.timestamp_depth) == (1000 + i, 2000 + i)
    out.close()
    # an aligned file is moved untouched; a missing input with the output present is skipped; unknown device: nothing to do
    st = calibrate.calibrate_sens(dst, str(scan / "moved.sens"), str(devdir / "structure_A.txt"), str(devdir / "structure_A.lut"))
    assert st["already_aligned"] == 1 and os.path.exists(scan / "moved.sens") and not os.path.exists(dst)
    st = calibrate.calibrate_sens(dst, str(scan / "moved.sens"), str(devdir / "structure_A.txt"), None)
    assert st["skipped_existing"] == 1
    with pytest.raises(Exception, match="no sens file"):
        calibrate.calibrate_sens(dst, str(scan / "nothing.sens"), str(devdir / "structure_A.txt"), None)
    (scan / "scan7.txt").write_text("deviceId = unknown-device\r\n")
    r = subprocess.run([exe, str(scan / "moved.sens"), dst, str(tmp_path / "devices.csv"), str(devdir)], capture_output=True, text=True)
    assert r.returncode == 0 and "no calibration name found" in r.stdout and os.path.exists(scan / "moved.sens")
