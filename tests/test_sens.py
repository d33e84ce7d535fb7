"""`.sens` codec, zlib and JPEG against the reference's own implementation (oracle/_ref/libref_sens.so, built from
/root/reference/SensReader/c++/src/sensorData.h) and against independent decoders (Python zlib, PIL).

CPU-only (`-m "not gpu"`): this is host logic of the hot path.
"""
import ctypes as C
import io
import os
import struct
import zlib

import numpy as np
import pytest

from scannet_amd import _abi, sens, synth


def _frames(n, W=64, H=48, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        d = (1500 + 40 * np.sin(np.arange(W * H) / 37.0 + i)).astype(np.uint16).reshape(H, W)
        d += rng.integers(0, 8, (H, W), dtype=np.uint16)
        d[rng.random((H, W)) < 0.03] = 0
        pose = synth.yaw_pose(1 + 0.1 * i, 2, 1.5, 0.05 * i)
        if i == 2:
            pose = np.full((4, 4), -np.inf, np.float32)  # tracking lost (sensorData.h:382)
        out.append((d, pose, rng.integers(0, 256, (H, W, 3), dtype=np.uint8)))
    return out


def test_zlib_roundtrip_and_interop():
    rng = np.random.default_rng(0)
    cases = [b"", b"a", b"abc" * 1000, bytes(100000), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
             (np.arange(307200, dtype=np.uint16) // 97 + 2000).tobytes()]
    for raw in cases:
        ours = sens.zlib_deflate(raw)
        assert zlib.decompress(ours) == raw                      # a stock inflater accepts our stream (incl. Adler-32)
        assert sens.zlib_inflate(ours, len(raw)) == raw
        for level in (0, 1, 6, 9):                               # stored, fixed and dynamic Huffman blocks
            assert sens.zlib_inflate(zlib.compress(raw, level), len(raw)) == raw
    # multi-block stream with a preset window crossing blocks
    co = zlib.compressobj(6)
    raw = b"".join([co.compress(cases[4][:30000]), co.flush(zlib.Z_FULL_FLUSH), co.compress(cases[2]), co.flush()])
    assert sens.zlib_inflate(raw, 200000) == cases[4][:30000] + cases[2]


def test_zlib_errors_and_adler_quirk():
    raw = b"hello hello hello hello" * 50
    z = bytearray(zlib.compress(raw))
    z[-1] ^= 0xFF  # corrupt Adler-32: the reference (stb) never checks it, neither do we
    assert sens.zlib_inflate(bytes(z), len(raw)) == raw
    with pytest.raises(_abi.ScanfuseError):
        sens.zlib_inflate(b"\x78\x9d" + bytes(z[2:]), len(raw))  # bad FCHECK
    with pytest.raises(_abi.ScanfuseError):
        sens.zlib_inflate(zlib.compress(raw), len(raw) - 1)      # output larger than the frame
    with pytest.raises(_abi.ScanfuseError):
        sens.zlib_inflate(zlib.compress(raw)[:20], len(raw))     # truncated
    with pytest.raises(_abi.ScanfuseError):
        sens.zlib_inflate(b"\x78\x9c\x07", 10)                    # reserved block type


def _write_ours(path, frames, W, H, raw_color=True):
    sd = sens.SensorData.create(W if raw_color else 0, H if raw_color else 0, W, H, synth.intrinsic_matrix(W, H), synth.intrinsic_matrix(W, H),
                                color_compression=0, depth_compression=1, sensor_name="StructureSensor")
    for i, (d, p, c) in enumerate(frames):
        sd.add_frame(d, p, color=c if raw_color else None, timestamp_color=1000 * i, timestamp_depth=1000 * i + 7)
    sd.save(path)
    return sd


def test_own_roundtrip(tmp_path):
    W, H = 64, 48
    frames = _frames(5, W, H)
    p = str(tmp_path / "a.sens")
    _write_ours(p, frames, W, H)
    sd = sens.SensorData(p)
    assert (sd.version, sd.sensor_name, sd.depth_width, sd.depth_height, sd.depth_shift) == (4, "StructureSensor", W, H, 1000.0)
    assert sd.depth_compression_type == "zlib_ushort" and sd.color_compression_type == "raw" and sd.num_frames == 5
    assert np.array_equal(sd.intrinsic_depth, synth.intrinsic_matrix(W, H))
    for i, (d, pose, c) in enumerate(frames):
        f = sd.frames[i]
        assert np.array_equal(f.decompress_depth(), d) and np.array_equal(f.decompress_color(), c)
        assert np.array_equal(f.camera_to_world, pose) and f.valid_pose == (i != 2)
        assert (f.timestamp_color, f.timestamp_depth) == (1000 * i, 1000 * i + 7)
    # byte-identical re-save of a loaded file, and pose rewrite (recons stage: s_overwriteOrigSensTrajectory)
    q = str(tmp_path / "b.sens")
    sd.save(q)
    assert open(p, "rb").read() == open(q, "rb").read()
    sd.set_pose(2, np.eye(4))
    sd.save(q)
    assert sens.SensorData(q).frames[2].valid_pose
    with pytest.raises(_abi.ScanfuseError):
        sd.frames[0]._o and _abi.check(_abi.lib().sf_sens_decode_depth(sd._h, 5, np.zeros(W * H, np.uint16).ctypes.data_as(C.c_void_p)))


def test_reference_reads_ours_and_we_read_reference(oracle, tmp_path):
    if not oracle.ref_sens_available():
        pytest.skip("oracle/_ref/libref_sens.so not built (needs /root/reference)")
    R = oracle.ref_sens()
    W, H = 64, 48
    frames = _frames(6, W, H, seed=11)
    ours = str(tmp_path / "ours.sens")
    _write_ours(ours, frames, W, H)
    # (1) the reference codec reads our file
    h = R.ref_sens_open(ours.encode())
    assert h
    info = oracle.RefSensInfo()
    R.ref_sens_get_info(h, C.byref(info))
    assert (info.version, info.depth_width, info.depth_height, info.num_frames, info.depth_compression, info.color_compression) == (4, W, H, 6, 1, 0)
    assert info.sensor_name == b"StructureSensor" and info.depth_shift == 1000.0
    assert np.array_equal(np.array(info.depth_intrinsic, np.float32).reshape(4, 4), synth.intrinsic_matrix(W, H))
    for i, (d, pose, c) in enumerate(frames):
        out = np.zeros((H, W), np.uint16)
        assert R.ref_sens_decode_depth(h, i, out.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(out, d)
        col = np.zeros((H, W, 3), np.uint8)
        assert R.ref_sens_decode_color(h, i, col.ctypes.data_as(C.c_void_p)) == 0 and np.array_equal(col, c)
        pm = np.zeros(16, np.float32)
        R.ref_sens_pose(h, i, pm.ctypes.data_as(C.c_void_p))
        assert np.array_equal(pm.reshape(4, 4), pose, equal_nan=True)
    R.ref_sens_close(h)
    # (2) we read a file written by the reference writer (stb deflate, single fixed-Huffman block)
    K = synth.intrinsic_matrix(W, H)
    w = R.ref_sens_create(W, H, W, H, K.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), 1000.0, b"StructureSensor")
    for i, (d, pose, c) in enumerate(frames):
        pc = np.ascontiguousarray(pose, np.float32)
        assert R.ref_sens_add_frame(w, c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), pc.ctypes.data_as(C.c_void_p), 5 * i, 5 * i + 1) == 0
    theirs = str(tmp_path / "theirs.sens")
    assert R.ref_sens_save(w, theirs.encode()) == 0
    R.ref_sens_close(w)
    sd = sens.SensorData(theirs)
    assert sd.num_frames == 6 and sd.sensor_name == "StructureSensor" and sd.num_imu_frames == 0
    for i, (d, pose, c) in enumerate(frames):
        f = sd.frames[i]
        assert np.array_equal(f.decompress_depth(), d) and np.array_equal(f.decompress_color(), c)
        assert np.array_equal(f.camera_to_world, pose, equal_nan=True)
        assert (f.timestamp_color, f.timestamp_depth) == (5 * i, 5 * i + 1)
    # header + frame records are laid out identically: only the deflate streams differ
    a, b = open(ours, "rb").read(), open(theirs, "rb").read()
    hdr = 4 + 8 + len("StructureSensor") + 256 + 8 + 16 + 4 + 8
    assert a[:hdr - 8] == b[:hdr - 8]


def test_python_struct_view_of_our_file(tmp_path):
    """Independent cross-check with the struct formats of SensReader/python/SensorData.py:14-20,54-74."""
    W, H = 32, 24
    frames = _frames(3, W, H)
    p = str(tmp_path / "s.sens")
    _write_ours(p, frames, W, H, raw_color=False)
    with open(p, "rb") as f:
        assert struct.unpack("I", f.read(4))[0] == 4
        n = struct.unpack("Q", f.read(8))[0]
        assert f.read(n) == b"StructureSensor"
        f.read(4 * 64)
        cc, dc, cw, ch, dw, dh = struct.unpack("iiIIII", f.read(24))
        assert (cc, dc, dw, dh) == (0, 1, W, H)
        assert struct.unpack("f", f.read(4))[0] == 1000.0
        assert struct.unpack("Q", f.read(8))[0] == 3
        for d, pose, _ in frames:
            m = np.frombuffer(f.read(64), np.float32).reshape(4, 4)
            assert np.array_equal(m, pose, equal_nan=True)
            _, _, cb, db = struct.unpack("QQQQ", f.read(32))
            assert cb == 0
            assert np.array_equal(np.frombuffer(zlib.decompress(f.read(db)), np.uint16).reshape(H, W), d)
        assert struct.unpack("Q", f.read(8))[0] == 0 and f.read() == b""


def test_bad_files(tmp_path):
    p = tmp_path / "bad.sens"
    p.write_bytes(struct.pack("I", 3) + bytes(100))
    with pytest.raises(_abi.ScanfuseError, match="version"):
        sens.SensorData(str(p))
    with pytest.raises(_abi.ScanfuseError, match="could not open"):
        sens.SensorData(str(tmp_path / "missing.sens"))
    W, H = 32, 24
    q = str(tmp_path / "ok.sens")
    _write_ours(q, _frames(2, W, H), W, H)
    data = open(q, "rb").read()
    (tmp_path / "trunc.sens").write_bytes(data[:len(data) // 2])
    with pytest.raises(_abi.ScanfuseError, match="truncated"):
        sens.SensorData(str(tmp_path / "trunc.sens"))


def _jpeg_bytes(img, quality, subsampling):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=quality, subsampling=subsampling)
    return buf.getvalue()


def _test_picture(W, H, seed=1, noise=20):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 3 % 256)], -1).astype(np.int32)
    img[H // 3:H // 2 + 1, W // 3:W // 2 + 1] = (200, 30, 60)
    return (img + rng.integers(-noise, noise + 1, img.shape)).clip(0, 255).astype(np.uint8)


def _decode_both(oracle, tmp_path, blob, W, H, color_compression=2):
    """The blob as the colour frame of a one-frame .sens, decoded by this library and by the reference's SensorData (stb_image)."""
    sd = sens.SensorData.create(W, H, 8, 8, np.eye(4), np.eye(4), color_compression=color_compression, depth_compression=0)
    sd.add_frame(np.zeros((8, 8), np.uint16), np.eye(4), color=blob)
    p = str(tmp_path / "c.sens")
    sd.save(p)
    ours = sens.SensorData(p).frames[0].decompress_color()
    R = oracle.ref_sens()
    h = R.ref_sens_open(p.encode())
    ref = np.zeros((H, W, 3), np.uint8)
    rc = R.ref_sens_decode_color(h, 0, ref.ctypes.data_as(C.c_void_p))
    R.ref_sens_close(h)
    assert rc == 0
    return ours, ref


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_jpeg_decode_against_pil(subsampling):
    """An independent decoder (libjpeg-turbo through PIL): T.81 defines no bit-exact IDCT, so this one is a tolerance."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from scannet_amd import calibrate
    W, H = 136, 104  # not a multiple of the 16x16 MCU
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 3 % 256)], -1).astype(np.uint8)
    img[30:60, 40:90] = (200, 30, 60)
    blob = _jpeg_bytes(img, 90, subsampling)
    ours = calibrate.jpeg_decode(blob, W, H).astype(np.int32)
    pil = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB")).astype(np.int32)
    diff = np.abs(ours - pil)
    # 4:2:2: the reference's (hence our) resampler weights the last chroma column differently from libjpeg (jpeg_idct.h)
    assert (diff.max(-1) > 4).mean() < 0.02 and diff.mean() < 0.6


@pytest.mark.parametrize("size", [(136, 104), (17, 9), (1, 1), (2, 3), (33, 31), (16, 16), (15, 16), (3, 50), (640, 480)])
def test_jpeg_decode_identical_to_reference(oracle, tmp_path, size):
    """the reference decodes colour with stb_image's integer IDCT / resampler / YCbCr (sensorData.h:609-616,
    stb_image.h:1969,2871-2933,3095) -- integer work, so the bar is identity: 4:4:4 / 4:2:2 / 4:2:0, sizes that are not whole MCUs,
    three qualities, with and without restart markers."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    if not oracle.ref_sens_available():
        pytest.skip("reference build absent")
    W, H = size
    img = _test_picture(W, H)
    for sub in (0, 1, 2):
        for q, rst in ((90, 0), (30, 3), (100, 0)) if W * H < 100000 else ((90, 0),):
            buf = io.BytesIO()
            kw = dict(restart_marker_blocks=rst) if rst else {}
            Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=sub, **kw)
            ours, ref = _decode_both(oracle, tmp_path, buf.getvalue(), W, H)
            assert np.array_equal(ours, ref), (size, sub, q, rst, int(np.abs(ours.astype(int) - ref).max()))


LAYOUTS = {"440": ((1, 2), (1, 1), (1, 1)), "411": ((4, 1), (1, 1), (1, 1)), "410": ((4, 2), (1, 1), (1, 1)), "422": ((2, 1), (1, 1), (1, 1)),
           "420": ((2, 2), (1, 1), (1, 1)), "v4": ((1, 4), (1, 1), (1, 1)), "h2v4": ((2, 4), (1, 1), (1, 1)), "luma-subsampled": ((1, 1), (2, 2), (2, 2)),
           "mixed": ((2, 2), (2, 1), (1, 2)), "h3": ((3, 1), (1, 1), (1, 1)), "grey": ((1, 1),)}


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_jpeg_unusual_layouts_identical_to_reference(oracle, tmp_path, layout):
    """Sampling factors PIL cannot write (tests/jpeg_tools.py): stb resamples (1,2) / (2,1) / (2,2) with the triangle filter and everything
    else with nearest neighbour (stb_image.h:3356-3361); grey pictures become r = g = b."""
    if not oracle.ref_sens_available():
        pytest.skip("reference build absent")
    from tests import jpeg_tools
    for W, H in ((40, 24), (17, 9), (1, 1), (33, 35)):
        img = _test_picture(W, H, seed=W, noise=30)
        if layout == "grey":
            img = img[..., 0]
        for rst in (0, 2):
            blob = jpeg_tools.encode(img, LAYOUTS[layout], qstep=3, restart=rst)
            ours, ref = _decode_both(oracle, tmp_path, blob, W, H)
            assert np.array_equal(ours, ref), (layout, W, H, rst)


def test_jpeg_rejects_what_the_reference_rejects():
    """Over-subscribed Huffman code lengths (stb_image.h:1543 'bad code lengths'; before this check a 230-byte file wrote 100 KB past
    the look-up table), 16-bit quantisation tables (stb_image.h:2625), a DRI segment that is not 4 bytes long (:2612), a stream
    that ends inside a run of 0xFF fill bytes."""
    from tests import jpeg_tools
    from scannet_amd import calibrate
    good = jpeg_tools.encode(_test_picture(16, 16), ((2, 2), (1, 1), (1, 1)))
    assert calibrate.jpeg_decode(good, 16, 16).shape == (16, 16, 3)
    i = good.index(b"\xff\xc4")
    bad = bytearray(good)
    bad[i + 5:i + 5 + 16] = bytes([3, 9] + [0] * 14)   # three codes of length 1 (and still 12 values: only the code-space check can object)
    with pytest.raises(_abi.ScanfuseError, match="code lengths"):
        calibrate.jpeg_decode(bytes(bad), 16, 16)
    j = good.index(b"\xff\xdb")
    bad = bytearray(good)
    bad[j + 4] = 0x10
    with pytest.raises(_abi.ScanfuseError, match="16-bit quantisation"):
        calibrate.jpeg_decode(bytes(bad), 16, 16)
    k = good.index(b"\xff\xda")
    with pytest.raises(_abi.ScanfuseError, match="DRI"):
        calibrate.jpeg_decode(good[:k] + b"\xff\xdd\x00\x02" + good[k:], 16, 16)
    with pytest.raises(_abi.ScanfuseError):
        calibrate.jpeg_decode(good[:k] + b"\xff\xff\xff", 16, 16)


def test_jpeg_restart_intervals_against_pil(tmp_path):
    """DRI / RSTn (T.81 B.2.4.4, E.2.4): the predictors restart, the bit reader realigns on the marker -- with and without chroma
    subsampling, an interval that does not divide the MCU count."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from scannet_amd import calibrate
    W, H = 136, 104
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 3 % 256)], -1).astype(np.uint8)
    img[30:60, 40:90] = (200, 30, 60)
    for sub, blocks in ((0, 3), (2, 5), (1, 1)):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="JPEG", quality=90, subsampling=sub, restart_marker_blocks=blocks)
        blob = buf.getvalue()
        assert b"\xff\xdd" in blob and any(bytes([0xFF, 0xD0 + k]) in blob for k in range(8))
        ours = calibrate.jpeg_decode(blob, W, H).astype(np.int32)
        pil = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB")).astype(np.int32)
        diff = np.abs(ours - pil)
        # not identity: libjpeg's IDCT differs, and at 4:2:2 so does its last chroma column (identity is against the reference, above)
        assert (diff.max(-1) > 4).mean() < 0.02 and diff.mean() < 0.6, (sub, blocks, diff.max())

