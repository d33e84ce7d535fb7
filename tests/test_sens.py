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


def _deflate_stream_with_15_bit_codes(align):
    """A dynamic-Huffman block whose literal/length and distance codes reach 15 bits (lengths 1..14, 15, 15: a complete code), shifted by
    `align` one-bit literals: literal + literal + length + 15-bit distance code + 13 extra bits = 63 bits between two refills."""
    # Use a bit writer (LSB first)
    bits=[]
    def put(v,n):
        for i in range(n): bits.append((v>>i)&1)
    def put_code(code, n):  # huffman codes MSB first
        for i in range(n-1,-1,-1): bits.append((code>>i)&1)
    # literal/length lengths: make a code with lengths: choose symbols: we need a complete (or valid) code with 15-bit codes.
    # lengths 1,2,...,14,15,15 is complete (Kraft sum = 1). assign to 16 symbols.
    ll = [0]*286
    syms = [65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 256, 97, 285]  # lengths 1..14, then 15,15 for 'a'(97) and 285 (len 258)
    lens = list(range(1,15)) + [15,15]
    for s_,l in zip(syms,lens): ll[s_]=l
    dl = [0]*30
    dsyms = list(range(14)) + [28, 29]
    for s_,l in zip(dsyms,lens): dl[s_]=l
    def canon(lengths):
        maxl=max(lengths); bl=[0]*(maxl+2)
        for l in lengths:
            if l: bl[l]+=1
        code=0; nxt=[0]*(maxl+2)
        for b in range(1,maxl+1):
            code=(code+bl[b-1])<<1; nxt[b]=code
        codes={}
        for i,l in enumerate(lengths):
            if l: codes[i]=(nxt[l],l); nxt[l]+=1
        return codes
    lc, dc = canon(ll), canon(dl)
    # code length alphabet: need symbols 0..15 and maybe repeat; simply give all 19 length 5 (32 slots >= 19: incomplete but is that accepted? use lengths: 16 symbols at 4 bits = complete)
    # use only symbols 0..15 with 4-bit codes each (complete), no repeats
    cl = [4]*16 + [0,0,0]
    cc = canon(cl)
    order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
    put(1,1); put(2,2)
    put(286-257,5); put(30-1,5); put(19-4,4)
    for o in order: put(cl[o],3)
    for l in ll+dl: put_code(*cc[l])
    # data: some short literals to shift alignment, then 'a' * many with 15-bit codes: literal 'a' (15 bits), literal 'a', then match len 258 (15 bits, sym 285 no extra) dist code 29 (15 bits) + 13 extra bits
    out=bytearray()
    for i in range(align):
        put_code(*lc[65]); out.append(65)
    # enough history for distance 24577+: fill with literal 'A' (1-bit code)
    for i in range(30000):
        put_code(*lc[65]); out.append(65)
    for rep in range(40):
        put_code(*lc[97]); out.append(97)
        put_code(*lc[97]); out.append(97)
        put_code(*lc[285]);  # len 258
        put_code(*dc[29]); ext = (rep*797+5) & 8191; put(ext,13)
        d = 24577+ext
        for k in range(258): out.append(out[-d])
        put_code(*lc[97]); out.append(97)
        put_code(*lc[77]); out.append(77)   # 13-bit
        put_code(*lc[285]); put_code(*dc[28]); ext=(rep*31)&8191; put(ext,13); d=16385+ext
        for k in range(258): out.append(out[-d])
    put_code(*lc[256])
    while len(bits)%8: bits.append(0)
    raw=bytes(sum(bits[i+j]<<j for j in range(8)) for i in range(0,len(bits),8))
    return raw, bytes(out)


def test_inflate_codes_of_maximum_length_at_every_alignment():
    """literal, second symbol, length extra bits, distance code and distance extra bits were decoded on one refill (56 bits
    guaranteed, 63 needed): wrong distances at some alignments.  Python's zlib reads the same streams."""
    for a in range(16):
        raw, want = _deflate_stream_with_15_bit_codes(a)
        assert zlib.decompressobj(-15).decompress(raw) == want
        z = b"\x78\x9c" + raw + zlib.adler32(want).to_bytes(4, "big")
        assert sens.zlib_inflate(z, len(want)) == want, a


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


def test_frame_blobs_are_stored_and_returned_as_given(oracle, tmp_path):
    """sf_sens_frame_blobs = RGBDFrame::getColorCompressed / getDepthCompressed (sensorData.h:418-429); sf_sens_add_frame_blobs stores blobs some other
    writer compressed (here: the reference's stb deflate, when it is built) without touching them: the file re-saved from them is the source file."""
    import zlib
    W, H = 64, 48
    frames = _frames(5, W, H, seed=3)
    src = str(tmp_path / "src.sens")
    if oracle.ref_sens_available() and hasattr(oracle.ref_sens(), "ref_sens_add_frames_mt"):
        oracle.ref_write_sens(src, np.stack([d for d, _, _ in frames]), np.stack([p for _, p, _ in frames]), synth.intrinsic_matrix(W, H), rgb=np.stack([c for _, _, c in frames]),
                              timestamp_step=5)
    else:
        _write_ours(src, frames, W, H)
    a = sens.SensorData(src)
    K = synth.intrinsic_matrix(W, H)
    b = sens.SensorData.create(W, H, W, H, K, K, sensor_name="StructureSensor")
    for i in range(5):
        f = a.frames[i]
        assert len(f.depth_compressed) == f.depth_size_bytes and len(f.color_compressed) == f.color_size_bytes == W * H * 3
        assert zlib.decompress(f.depth_compressed) == frames[i][0].tobytes() and f.color_compressed == frames[i][2].tobytes()
        b.add_frame_blobs(f.depth_compressed, f.camera_to_world, color_blob=f.color_compressed, timestamp_color=f.timestamp_color, timestamp_depth=f.timestamp_depth)
    dst = str(tmp_path / "dst.sens")
    b.save(dst)
    assert open(src, "rb").read() == open(dst, "rb").read()
    with pytest.raises(_abi.ScanfuseError):
        b.add_frame_blobs(b"", np.eye(4), color_blob=b"abc")      # a raw colour frame of the wrong size


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


@pytest.mark.parametrize("size", [(136, 104), (17, 9), (1, 1), (2, 3), (33, 31), (16, 16), (15, 16), (3, 50), (640, 480)])
def test_progressive_jpeg_identical_to_reference(oracle, tmp_path, size):
    """The reference's decoder takes progressive pictures too (stb_image.h:1771-1900, 2520-2556, 2582-2598) and RGBDFrame::decompressColorAlloc_stb hands
    them through: DC scans interleaved, AC bands per component, successive approximation with refinement scans, restart intervals, tables redefined
    between scans -- libjpeg's progression as PIL writes it, 4:4:4 / 4:2:2 / 4:2:0, sizes that are not whole MCUs; the same bytes as the reference."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    if not oracle.ref_sens_available():
        pytest.skip("reference build absent")
    W, H = size
    img = _test_picture(W, H)
    n = 0
    for sub in (0, 1, 2):
        for q, rst, opt in ((90, 0, False), (30, 3, True), (100, 0, True), (5, 0, False)) if W * H < 100000 else ((90, 0, True),):
            buf = io.BytesIO()
            kw = dict(restart_marker_blocks=rst) if rst else {}
            Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=sub, progressive=True, optimize=opt, **kw)
            blob = buf.getvalue()
            assert b"\xff\xc2" in blob and blob.count(b"\xff\xda") > 3            # SOF2 and several scans
            ours, ref = _decode_both(oracle, tmp_path, blob, W, H)
            assert np.array_equal(ours, ref), (size, sub, q, rst, int(np.abs(ours.astype(int) - ref).max()))
            n += 1
    assert n >= 3
    grey = io.BytesIO()
    Image.fromarray(img[..., 1]).save(grey, format="JPEG", quality=80, progressive=True)
    ours, ref = _decode_both(oracle, tmp_path, grey.getvalue(), W, H)
    assert np.array_equal(ours, ref)


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


def _png_bytes(arr, ctype, depth=8, interlace=False, palette=None, filter_type=None):
    """Minimal PNG writer for the tests: arr [H, W] or [H, W, C] of samples (unscaled), any colour type / bit depth, optional Adam7,
    one scan-line filter type per image (None = 0), sub-byte depths packed MSB first."""
    H, W = arr.shape[:2]
    a = arr.reshape(H, W, -1).astype(np.uint8)
    ch = a.shape[2]

    def rows(sub):
        h, w = sub.shape[:2]
        if depth == 8:
            packed = sub.reshape(h, w * ch)
        else:
            bits = np.unpackbits(sub.reshape(h, w, 1), axis=2)[:, :, 8 - depth:].reshape(h, w * depth)
            bits = np.pad(bits, ((0, 0), (0, (-bits.shape[1]) % 8)))
            packed = np.packbits(bits, axis=1)
        bpp = max(1, ch * depth // 8)
        out = bytearray()
        prev = np.zeros(packed.shape[1], np.int32)
        for r in packed.astype(np.int32):
            ft = filter_type or 0
            left = np.concatenate([np.zeros(bpp, np.int32), r[:-bpp]]) if bpp <= len(r) else np.zeros_like(r)
            ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if bpp <= len(r) else np.zeros_like(r)
            if ft == 0: f = r
            elif ft == 1: f = r - left
            elif ft == 2: f = r - prev
            elif ft == 3: f = r - ((left + prev) >> 1)
            else:
                pp = left + prev - ul
                pa, pb, pc = abs(pp - left), abs(pp - prev), abs(pp - ul)
                f = r - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            out += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
            prev = r
        return bytes(out)

    if interlace:
        raw = b""
        for x0, y0, xs, ys in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = a[y0::ys, x0::xs]
            if sub.size:
                raw += rows(sub)
    else:
        raw = rows(a)

    def chunk(t, body):
        return len(body).to_bytes(4, "big") + t + body + zlib.crc32(t + body).to_bytes(4, "big")
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", W.to_bytes(4, "big") + H.to_bytes(4, "big") + bytes([depth, ctype, 0, 0, 1 if interlace else 0]))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    return out + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def test_png_colour_identical_to_reference(oracle, tmp_path):
    """TYPE_PNG colour frames go through the same stb call as JPEG in the reference (sensorData.h:346-351,609-616): 8-bit grey, grey +
    alpha, RGB, RGBA and palette images, every scan-line filter, with and without Adam7 -- identical to the reference's decoder and to PIL."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    if not oracle.ref_sens_available():
        pytest.skip("reference build absent")
    rng = np.random.default_rng(4)
    n = 0
    for W, H in ((37, 21), (1, 1), (8, 8), (5, 3), (64, 48)):
        for ctype, ch in ((0, 1), (4, 2), (2, 3), (6, 4), (3, 1)):
            for interlace in (False, True):
                for ft in (0, 1, 2, 3, 4):
                    arr = rng.integers(0, 256, (H, W, ch), dtype=np.uint8)
                    arr[:H // 2] = (arr[:H // 2] // 64) * 64
                    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8) if ctype == 3 else None
                    blob = _png_bytes(arr, ctype, 8, interlace, pal, ft)
                    ours, ref = _decode_both(oracle, tmp_path, blob, W, H, color_compression=1)
                    pil = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
                    assert np.array_equal(ours, ref) and np.array_equal(ours, pil), (W, H, ctype, interlace, ft)
                    n += 1
    assert n == 250


def test_png_colour_sub_byte_depths_follow_the_png_specification(tmp_path):
    """1 / 2 / 4-bit grey and palette PNGs.  The reference's stb_image v2.08 un-filters such rows against UNINITIALISED memory (its `prior`
    pointer is taken before the row is right-aligned for in-place expansion, stb_image.h:4004-4012), so its output for filters 2-4 is not a
    function of the file; here the PNG specification decides, checked against PIL.  Grey is scaled to 0..255 as stb and PIL both do."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(5)
    for W, H in ((37, 21), (1, 1), (8, 8), (5, 3), (16, 4)):
        for depth in (1, 2, 4):
            for ctype in (0, 3):
                for interlace in (False, True):
                    for ft in (0, 2, 4):
                        arr = rng.integers(0, 1 << depth, (H, W), dtype=np.uint8)
                        pal = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8) if ctype == 3 else None
                        blob = _png_bytes(arr, ctype, depth, interlace, pal, ft)
                        sd = sens.SensorData.create(W, H, 8, 8, np.eye(4), np.eye(4), color_compression=1, depth_compression=0)
                        sd.add_frame(np.zeros((8, 8), np.uint16), np.eye(4), color=blob)
                        p = str(tmp_path / "p.sens")
                        sd.save(p)
                        ours = sens.SensorData(p).frames[0].decompress_color()
                        pil = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
                        assert np.array_equal(ours, pil), (W, H, depth, ctype, interlace, ft)
    # 16-bit samples: the reference refuses them ("1/2/4/8-bit only"), so does the .sens reader
    blob16 = _png_bytes(np.zeros((4, 4, 2), np.uint8), 0, 8)   # patched below into a 16-bit grey header
    blob16 = blob16[:24] + bytes([16]) + blob16[25:]
    sd = sens.SensorData.create(4, 4, 8, 8, np.eye(4), np.eye(4), color_compression=1, depth_compression=0)
    sd.add_frame(np.zeros((8, 8), np.uint16), np.eye(4), color=blob16)
    p = str(tmp_path / "p16.sens")
    sd.save(p)
    with pytest.raises(_abi.ScanfuseError):
        sens.SensorData(p).frames[0].decompress_color()
    # a 60-byte frame that announces 30000 x 30000 RGBA: refused on the IHDR (size mismatch with the .sens header), before the 3.6 GB its
    # rows would take are allocated -- and a frame of the RIGHT size whose IDAT cannot possibly hold it is refused before the buffer exists
    import resource
    tiny = _png_bytes(np.zeros((4, 4), np.uint8), 0, 8)
    hostile = tiny[:16] + (30000).to_bytes(4, "big") + (30000).to_bytes(4, "big") + bytes([8, 6]) + tiny[26:]
    for (w, h, blob, what) in ((4, 4, hostile, "header says"), (30000, 30000, hostile, "compressed bytes present")):
        sd = sens.SensorData.create(w, h, 8, 8, np.eye(4), np.eye(4), color_compression=1, depth_compression=0)
        sd.add_frame(np.zeros((8, 8), np.uint16), np.eye(4), color=blob)
        p = str(tmp_path / "hostile.sens")
        sd.save(p)
        before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        with pytest.raises(_abi.ScanfuseError, match=what):
            sens.SensorData(p).frames[0].decompress_color()
        if not os.environ.get("SCANFUSE_LIBRARY"):   # (the sanitizer build keeps shadow memory and quarantines: its RSS says nothing)
            assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 200 * 1024   # KiB: nothing image-sized was touched


def test_occipital_depth_frames_in_a_sens(oracle, tmp_path):
    """TYPE_OCCI_USHORT (sensorData.h:672-684,711-722): the writer codes the values as given, the reader decodes and maps shift -> mm.
    Checked against the reference's own uplinksimple headers (oracle/_ref/libref_occ.so): same stream bytes in, same millimetres out."""
    W, H = 64, 48
    rng = np.random.default_rng(6)
    shifts = (600 + 200 * np.sin(np.arange(W * H) / 50.0)).astype(np.uint16).reshape(H, W)
    shifts[rng.random((H, W)) < 0.05] = 0
    shifts[rng.random((H, W)) < 0.02] = 2047
    K = synth.intrinsic_matrix(W, H)
    sd = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=2)
    sd.add_frame(shifts, np.eye(4))
    p = str(tmp_path / "o.sens")
    sd.save(p)
    got = sens.SensorData(p).frames[0].decompress_depth()
    L = _abi.lib()
    want = shifts.copy().reshape(-1)
    L.sf_occ_shift2depth_buffer.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    _abi.check(L.sf_occ_shift2depth_buffer(want.ctypes.data, want.size, 0))
    assert np.array_equal(got.reshape(-1), want)
    ref_so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_occ.so")
    if os.path.exists(ref_so):
        R = C.CDLL(ref_so)
        R.ref_occ_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        R.ref_occ_shift2depth.argtypes = [C.c_uint16]
        R.ref_occ_shift2depth.restype = C.c_uint16
        data = open(p, "rb").read()
        nbytes = sens.SensorData(p).frames[0].depth_size_bytes
        stream = np.frombuffer(data[-8 - nbytes:-8] + bytes(8), np.uint8)   # the frame's depth blob (+ padding: the reference reads a little past the last code)
        ref = np.zeros(W * H, np.uint16)
        R.ref_occ_decode(stream.ctypes.data, nbytes, W * H, ref.ctypes.data)
        ref = np.array([R.ref_occ_shift2depth(int(v)) for v in ref], np.uint16)   # uplinksimple::shift2depth(buffer, n) is this loop
        assert np.array_equal(ref, got.reshape(-1))


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



def test_bulk_depth_writer_writes_the_file_of_the_frame_by_frame_writer(tmp_path):
    """sf_sens_add_depth_frames (threaded deflate) against n sf_sens_add_frame calls: byte-identical files, zlib and raw depth."""
    W, H, N = 64, 48, 37
    rng = np.random.default_rng(21)
    depth = rng.integers(0, 5000, (N, H, W), dtype=np.uint16)
    depth[3] = 0
    poses = rng.normal(size=(N, 4, 4)).astype(np.float32)
    K = synth.intrinsic_matrix(W, H)
    for comp in (1, 0):
        a = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=comp)
        for i in range(N):
            a.add_frame(depth[i], poses[i], timestamp_depth=1000 + 33333 * i)
        b = sens.SensorData.create(0, 0, W, H, K, K, depth_compression=comp)
        b.add_depth_frames(depth[:20], poses[:20], timestamp0=1000, threads=5)
        b.add_depth_frames(depth[20:], poses[20:], timestamp0=1000 + 33333 * 20, threads=0)
        pa, pb = str(tmp_path / ("a%d.sens" % comp)), str(tmp_path / ("b%d.sens" % comp))
        a.save(pa)
        b.save(pb)
        assert open(pa, "rb").read() == open(pb, "rb").read()
        r = sens.SensorData(pb)
        assert r.num_frames == N and np.array_equal(r.frames[36].decompress_depth(), depth[36])


def test_imu_frames_and_the_closest_one_to_a_frame(oracle, tmp_path):
    """m_IMUFrames read back as stored, and findClosestIMUFrame(frameIdx, basedOnRGB) (sensorData.h:1000-1044) against the reference itself on the file
    both read: keys before / inside / after the recorded span, exact hits, ties between two neighbours (the later one wins), colour and depth stamps."""
    import ctypes as C
    rng = np.random.default_rng(9)
    W, H = 16, 12
    K = synth.intrinsic_matrix(W, H)
    sd = sens.SensorData.create(0, 0, W, H, K, K, sensor_name="StructureSensor")
    stamps = np.sort(rng.choice(np.arange(1000, 90000, 10), 60, replace=False)).astype(np.uint64)
    recs = np.zeros(60, sens.SensorData.IMU_DTYPE)
    for name in ("rotationRate", "acceleration", "magneticField", "attitude", "gravity"):
        recs[name] = rng.standard_normal((60, 3))
    recs["timeStamp"] = stamps
    keys = [0, 999, 1000, int(stamps[7]), int(stamps[7]) + 1, int(stamps[20] + stamps[21]) // 2, int(stamps[30]) - 1, int(stamps[58]) + 3, int(stamps[59]) - 1,
            int(stamps[59]) + 1, 10 ** 9] + [int(x) for x in rng.integers(0, 95000, 80)]
    keys += [int(stamps[i] + (stamps[i + 1] - stamps[i]) // 2) for i in (3, 11, 40)]          # exact midpoints (the gaps are multiples of 10): ties
    for i, k in enumerate(keys):
        sd.add_frame(np.full((H, W), 1000, np.uint16), np.eye(4, dtype=np.float32), timestamp_color=k, timestamp_depth=keys[-1 - i])
    for r in recs:
        sd.add_imu_frame(r)
    path = str(tmp_path / "imu.sens")
    sd.save(path)
    sd.close()
    sd = sens.SensorData(path)
    assert sd.num_imu_frames == 60 and sd.imu_frames.tobytes() == recs.tobytes()
    R = oracle.ref_sens() if oracle.ref_sens_available() else None
    h = R.ref_sens_open(path.encode()) if R is not None and hasattr(R, "ref_sens_find_closest_imu") else None
    checked = 0
    for f in range(len(keys)):
        for rgb in (True, False):
            key = keys[f] if rgb else keys[-1 - f]
            idx, rec = sd.find_closest_imu_frame(f, rgb)
            assert rec.tobytes() == recs[idx].tobytes()
            d = np.abs(stamps.astype(np.int64) - key)
            assert d[idx] == d.min()                                   # nearest in time ...
            if (d == d.min()).sum() == 2 and stamps[0] <= key <= stamps[-1]:
                assert idx == np.flatnonzero(d == d.min())[1]          # ... and of two equally near ones the later (`<` at :1035)
            if h is not None and key != int(stamps[-1]):               # at the last stamp the reference reads m_IMUFrames[size] (see sf_sens_find_closest_imu)
                assert R.ref_sens_find_closest_imu(h, f, 1 if rgb else 0) == idx, (f, rgb, key)
                checked += 1
    if h is not None:
        assert checked > 150
        R.ref_sens_close(h)
    no_imu = sens.SensorData.create(0, 0, W, H, K, K)
    no_imu.add_frame(np.zeros((H, W), np.uint16))
    with pytest.raises(Exception, match="no imu data available"):      # the reference's message (:1002)
        no_imu.find_closest_imu_frame(0)
    with pytest.raises(Exception, match="out of bounds"):
        sd.find_closest_imu_frame(10 ** 6)


def test_replace_append_and_equality_against_the_reference(oracle, tmp_path):
    """SensorData::replaceDepth / replaceColor / append / operator== (sensorData.h:948-964,1605-1650): the same edits made by this library and by the
    reference on the same files -- what each reads back of the other's result, the reference's operator== on the pairs, and the saved bytes of an append."""
    rng = np.random.default_rng(12)
    W, H = 24, 18
    K = synth.intrinsic_matrix(W, H)

    def scan(path, n, seed, lost=()):
        r = np.random.default_rng(seed)
        sd = sens.SensorData.create(W, H, W, H, K, K, color_compression=0, depth_compression=1, sensor_name="StructureSensor")
        for i in range(n):
            pose = np.full((4, 4), -np.inf, np.float32) if i in lost else synth.trajectory_pose(i * 31 + seed, 1200)
            sd.add_frame(r.integers(0, 5000, (H, W), dtype=np.uint16), pose, color=r.integers(0, 256, (H, W, 3), dtype=np.uint8), timestamp_color=100 + i, timestamp_depth=200 + i)
        sd.save(path)
        sd.close()
    a_path, b_path = str(tmp_path / "a.sens"), str(tmp_path / "b.sens")
    scan(a_path, 5, 1, lost=(2,))
    scan(b_path, 3, 2)
    a, a2, b = sens.SensorData(a_path), sens.SensorData(a_path), sens.SensorData(b_path)
    assert a == a2 and not (a == b) and a != b                       # a frame with the all -inf pose equals itself
    R = oracle.ref_sens() if oracle.ref_sens_available() and hasattr(oracle.ref_sens(), "ref_sens_equal") else None
    if R is not None:
        ra, ra2, rb = (R.ref_sens_open(p.encode()) for p in (a_path, a_path, b_path))
        assert R.ref_sens_equal(ra, ra2) == 1 and R.ref_sens_equal(ra, rb) == 0
    # replaceDepth on an OPENED file (the Calibrate stage's use): new pixels, depth time stamp zeroed, everything else kept
    new_d = rng.integers(0, 5000, (H, W), dtype=np.uint16)
    before = [(f.color_compressed, f.camera_to_world.copy(), f.timestamp_color) for f in a.frames]
    a.replace_depth(3, new_d)
    assert not (a == a2)
    f3 = a.frames[3]
    assert np.array_equal(f3.decompress_depth(), new_d) and f3.timestamp_depth == 0 and f3.timestamp_color == before[3][2] and f3.color_compressed == before[3][0]
    assert np.array_equal(a.frames[2].decompress_depth(), a2.frames[2].decompress_depth())
    edited = str(tmp_path / "edited.sens")
    a.save(edited)
    if R is not None:
        assert R.ref_sens_replace_depth(ra, 3, new_d.ctypes.data) == 0
        ref_edited = str(tmp_path / "ref_edited.sens")
        assert R.ref_sens_save(ra, ref_edited.encode()) == 0
        ours, theirs = sens.SensorData(edited), sens.SensorData(ref_edited)        # the depth streams differ (two deflaters), what they hold does not
        for fo, ft in zip(ours.frames, theirs.frames):
            assert np.array_equal(fo.decompress_depth(), ft.decompress_depth()) and fo.color_compressed == ft.color_compressed
            assert (fo.timestamp_color, fo.timestamp_depth) == (ft.timestamp_color, ft.timestamp_depth) and np.array_equal(fo.camera_to_world, ft.camera_to_world)
        r_ours = R.ref_sens_open(edited.encode())
        out = np.zeros((H, W), np.uint16)
        R.ref_sens_decode_depth(r_ours, 3, out.ctypes.data)
        assert np.array_equal(out, new_d)                                           # the reference reads the edited file
        R.ref_sens_close(r_ours)
    # replaceColor: raw pixels, colour time stamp zeroed
    new_c = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    a.replace_color(0, new_c)
    assert np.array_equal(a.frames[0].decompress_color(), new_c) and a.frames[0].timestamp_color == 0 and a.frames[0].timestamp_depth == 200
    with pytest.raises(Exception, match="colorWidth"):
        a.replace_color(0, new_c[:-1])
    with pytest.raises(Exception, match="out of bounds"):
        a.replace_depth(99, new_d)
    # append: b's frames behind a2's; the saved file is the reference's, byte for byte (blobs are copied as they are)
    a2.append(b)
    assert a2.num_frames == 8 and np.array_equal(a2.frames[6].decompress_depth(), b.frames[1].decompress_depth())
    joined = str(tmp_path / "joined.sens")
    a2.save(joined)
    if R is not None:
        assert R.ref_sens_append(ra2, rb) == 0
        ref_joined = str(tmp_path / "ref_joined.sens")
        assert R.ref_sens_save(ra2, ref_joined.encode()) == 0
        assert open(joined, "rb").read() == open(ref_joined, "rb").read()
        rj = R.ref_sens_open(joined.encode())
        assert R.ref_sens_equal(rj, ra2) == 1
        for h in (ra, ra2, rb, rj):
            R.ref_sens_close(h)
    other = sens.SensorData.create(W, H, W + 8, H, K, K, color_compression=0, depth_compression=1)
    with pytest.raises(Exception, match="incompatible"):
        a2.append(other)
    a2.append(a2)                                                                   # a file behind itself
    assert a2.num_frames == 16 and a2.frames[15].depth_compressed == a2.frames[7].depth_compressed


def test_apply_transform_moves_tracked_poses_only(tmp_path):
    W, H = 8, 6
    K = synth.intrinsic_matrix(W, H)
    sd = sens.SensorData.create(0, 0, W, H, K, K)
    poses = [synth.trajectory_pose(i * 40, 1200) for i in range(4)]
    poses[2] = np.full((4, 4), -np.inf, np.float32)
    for p in poses:
        sd.add_frame(np.zeros((H, W), np.uint16), p)
    t = synth.trajectory_pose(333, 1200)
    t[:3, 3] += np.float32(0.25)
    sd.apply_transform(t)
    for i, p in enumerate(poses):
        got = sd.frames[i].camera_to_world
        if i == 2:
            assert np.all(np.isneginf(got))
        else:
            assert np.allclose(got, t.astype(np.float64) @ p.astype(np.float64), atol=2e-6) and got.dtype == np.float32
    sd.save(str(tmp_path / "moved.sens"))
    assert np.array_equal(sens.SensorData(str(tmp_path / "moved.sens")).frames[1].camera_to_world, sd.frames[1].camera_to_world)


def test_streaming_writer_writes_the_in_memory_writers_file(oracle, tmp_path):
    """SensorData::LiveSensorDataWriter (sensorData.h:1112-1246): frames through the queue and the background thread (a cache of two frames: the calls block)
    = the file create + add_frame + save write, byte for byte; the reference's reader reads it; the numeric-suffix rule; errors of a frame reach the caller."""
    rng = np.random.default_rng(21)
    W, H, n = 40, 30, 25
    K = synth.intrinsic_matrix(W, H)
    depth = rng.integers(0, 6000, (n, H, W), dtype=np.uint16)
    colour = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    poses = [synth.trajectory_pose(i * 17, 1200) if i != 6 else np.full((4, 4), -np.inf, np.float32) for i in range(n)]
    whole = sens.SensorData.create(W, H, W, H, K, K, color_compression=0, depth_compression=1, sensor_name="StructureSensor")
    for i in range(n):
        whole.add_frame(depth[i], poses[i], color=colour[i], timestamp_color=10 * i, timestamp_depth=10 * i + 3)
    a = str(tmp_path / "whole.sens")
    whole.save(a)
    b = str(tmp_path / "stream.sens")
    with sens.SensorDataWriter(b, W, H, W, H, K, K, color_compression=0, depth_compression=1, sensor_name="StructureSensor", cache_frames=2) as w:
        for i in range(n):
            w.add_frame(depth[i], poses[i], color=colour[i], timestamp_color=10 * i, timestamp_depth=10 * i + 3)
    assert w.frames_written == n and w.path == b and open(a, "rb").read() == open(b, "rb").read()
    if oracle.ref_sens_available():
        R = oracle.ref_sens()
        h = R.ref_sens_open(b.encode())
        info = oracle.RefSensInfo()
        R.ref_sens_get_info(h, C.byref(info))
        out = np.zeros((H, W), np.uint16)
        R.ref_sens_decode_depth(h, n - 1, out.ctypes.data)
        assert info.num_frames == n and info.num_imu == 0 and np.array_equal(out, depth[n - 1])
        R.ref_sens_close(h)
    # overwrite=False: the name's numeric suffix counts up past the files that exist (:1118-1131)
    names = []
    for _ in range(3):
        with sens.SensorDataWriter(b, 0, 0, W, H, K, K, overwrite=False) as w2:
            w2.add_frame(depth[0])
        names.append(os.path.basename(w2.path))
    assert names == ["stream1.sens", "stream2.sens", "stream3.sens"] and sens.SensorData(str(tmp_path / "stream2.sens")).num_frames == 1
    assert open(b, "rb").read() == open(a, "rb").read()                      # and the first file was left alone
    # blobs as they are (transcoding): the frames of `whole` streamed into a third file
    c = str(tmp_path / "copy.sens")
    with sens.SensorDataWriter(c, W, H, W, H, K, K, color_compression=0, depth_compression=1, sensor_name="StructureSensor") as w3:
        for f in whole.frames:
            w3.add_frame_blobs(f.depth_compressed, f.camera_to_world, color_blob=f.color_compressed, timestamp_color=f.timestamp_color, timestamp_depth=f.timestamp_depth)
    assert open(c, "rb").read() == open(a, "rb").read()
    # a bad argument is refused at once; a frame that fails on the background thread (an Occipital shift value above 2047) fails the next call or the close
    with pytest.raises(Exception, match="colorWidth"):
        with sens.SensorDataWriter(str(tmp_path / "bad.sens"), W, H, W, H, K, K) as w4:
            w4.add_frame(depth[0], color=colour[0][:-1])
    w5 = sens.SensorDataWriter(str(tmp_path / "occ.sens"), 0, 0, W, H, K, K, depth_compression=2)
    w5.add_frame(np.full((H, W), 1000, np.uint16))
    w5.add_frame(np.full((H, W), 5000, np.uint16))                           # not a shift value
    seen = []
    try:
        for _ in range(50):
            w5.add_frame(np.full((H, W), 1000, np.uint16))
    except Exception as e:                                                   # the next add after the background thread met the frame ...
        seen.append(str(e))
    try:
        w5.close()
    except Exception as e:                                                   # ... and the close, which still finishes the file
        seen.append(str(e))
    assert seen and all("does not fit" in m for m in seen)
    assert sens.SensorData(str(tmp_path / "occ.sens")).num_frames == 1      # what was written before the failure is a valid file
    with pytest.raises(Exception, match="Unable to open"):
        sens.SensorDataWriter(str(tmp_path / "no" / "dir" / "x.sens"), 0, 0, W, H, K, K)
