"""The lane programs of the device's inflate (scannet_amd/csrc/inflate_lanes.h) on the HOST: tools/inflate_parallelism/emulate_gpu.cpp runs the
token kernel's 1024 lanes and the copy kernel's 64 lanes in lock step and writes what the device would; this file checks it against zlib.  The
GPU runs the same header (tests/test_gpu_pipeline.py checks the bytes on the device)."""
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

from tests import deflate_tools as dt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emulate(tmp_path_factory):
    d = tmp_path_factory.mktemp("inflate_lanes")
    exe = str(d / "emulate_gpu")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "scannet_amd", "csrc"), os.path.join(ROOT, "tools", "inflate_parallelism", "emulate_gpu.cpp"), "-o", exe],
                   check=True)

    def run(stream, expect):
        p, o = str(d / "s.z"), str(d / "s.out")
        open(p, "wb").write(stream)
        if os.path.exists(o):
            os.remove(o)
        r = subprocess.run([exe, p, str(expect), o], capture_output=True, text=True)
        return r.returncode, r.stdout, (open(o, "rb").read() if r.returncode == 0 else None)
    return run


def token_cases():
    """name -> token list: what a match finder would not give us on demand"""
    rng = np.random.default_rng(3)
    cases = {}
    cases["run of one byte"] = [7] + [(258, 1)] * 40 + [(3, 1), 9, 9, 9]                           # every match repeats the byte in front of it
    cases["run of a pair, odd lengths"] = [1, 2] + [(int(l), 2) for l in rng.integers(3, 259, 300)]
    t = [int(v) for v in rng.integers(0, 256, 8)]
    for i in range(4000):                                                                           # chains of short near matches: each copies what the last one wrote
        t.append((int(rng.integers(3, 7)), int(rng.integers(1, 9))))
        if i % 5 == 0:
            t.append(int(rng.integers(0, 256)))
    cases["chains of short near matches"] = t
    t = [int(v) for v in rng.integers(0, 256, 32768)]
    t += [(258, 32768)] * 100 + [(3, 32768), (258, 32767), (17, 24577), (258, 1)]
    cases["the longest distance"] = t
    t = [int(v) for v in rng.integers(0, 256, 1300)]
    for i in range(6000):                                                                           # distances around the group size and its multiples
        t.append((int(rng.integers(3, 40)), int(rng.choice([255, 256, 257, 511, 512, 513, 1279, 1280, 1281, 4, 3]))))
    cases["distances around 256"] = t
    cases["literals only"] = [int(v) for v in rng.integers(0, 256, 5000)]
    cases["tiny"] = [5, 6, 7, (9, 3)]
    for k in range(12):                                                                             # random mixes: every length, distances of every size class, literal runs
        r = np.random.default_rng(100 + k)
        t = [int(v) for v in r.integers(0, 256, int(r.integers(1, 400)))]
        produced = len(t)
        for _ in range(int(r.integers(200, 9000))):
            if r.random() < (0.1, 0.5, 0.9)[k % 3]:
                t.append(int(r.integers(0, 256)))
                produced += 1
            else:
                length = int(r.integers(3, 259)) if r.random() < 0.3 else int(r.integers(3, 12))
                top = min(produced, 32768)
                dist = int(r.integers(1, top + 1)) if r.random() < 0.5 else int(min(top, 2 ** int(r.integers(0, 16)) + int(r.integers(0, 3))))
                t.append((length, max(1, dist)))
                produced += length
        cases["random mix %d" % k] = t
    return cases


def test_token_streams_inflate_as_zlib_says(emulate):
    for name, tokens in token_cases().items():
        want = dt.apply_tokens(tokens)
        pad = (-len(want)) % 4
        tokens = tokens + [0] * pad                       # the device inflates to multiples of 4 bytes (depth frames are)
        want += bytes(pad)
        z = dt.zlib_stream(tokens)
        assert zlib.decompress(z) == want, name           # the helper against zlib
        rc, text, got = emulate(z, len(want))
        assert rc == 0, (name, text)
        assert got == want, name


def test_depth_frames_through_this_library_writer(emulate):
    """The streams the pipeline meets: this library's writer (one fixed block, as stb's) on depth-like frames, noise, constants."""
    from scannet_amd import sens, synth
    rng = np.random.default_rng(0)
    frames = [synth.render_room_depth(synth.trajectory_pose(37, 1200), 320, 240, noise_frame=3).tobytes(), bytes(153600),
              rng.integers(0, 65536, 76800, dtype=np.uint16).tobytes(), (np.arange(76800, dtype=np.uint16) // 7).tobytes(), b"\x01\x02\x03\x04" * 25]
    seen = []
    for raw in frames:
        z = sens.zlib_deflate(raw)
        assert zlib.decompress(z) == raw
        rc, text, got = emulate(z, len(raw))
        assert rc == 0 and got == raw, text
        m = re.search(r"chunks (\d+) \| tokens: (\d+) rounds", text)
        seen.append((int(m.group(1)), int(m.group(2))))
    assert max(c for c, _ in seen) > 100 and all(r <= 6 for _, r in seen), seen


def test_depth_frames_through_the_reference_writer(emulate, oracle):
    """The streams REAL .sens files hold: the reference's writer (SensorData::createFrame -> compressDepth -> stb::stbi_zlib_compress at quality 8,
    sensorData.h:659-670 / stb_image_write.h:721-823: one fixed-Huffman block, stb's hash-chain matcher -- other matches than this library's
    writer finds, 9 % more bytes) on furnished 640x480 frames, noise, constants and a ramp, through the lane programs on the host."""
    if not oracle.ref_sens_available() or not hasattr(oracle.ref_sens(), "ref_sens_add_frames_mt"):
        pytest.skip("oracle/_ref/libref_sens.so not built (needs /root/reference)")
    from scannet_amd import sens, synth
    rng = np.random.default_rng(5)
    W, H = 640, 480
    frames = [synth.render_room_depth(synth.trajectory_pose(211 * k, 1200), W, H, noise_frame=k, noise=2, boxes=synth.clutter_boxes()) for k in range(3)]
    frames += [np.zeros((H, W), np.uint16), np.full((H, W), 2000, np.uint16), rng.integers(0, 65536, (H, W), dtype=np.uint16), (np.arange(W * H, dtype=np.uint32) // 7).astype(np.uint16).reshape(H, W)]
    poses = np.stack([np.eye(4, dtype=np.float32)] * len(frames))
    blobs = oracle.ref_write_sens(None, np.stack(frames), poses, synth.intrinsic_matrix(W, H), want_blobs=True)
    seen = []
    for raw, z in zip(frames, blobs):
        raw = raw.tobytes()
        assert zlib.decompress(z) == raw and (z[2] & 7) == 3          # one final block, fixed code
        assert z != sens.zlib_deflate(raw) or len(set(raw)) <= 2        # not this library's stream (the constants may coincide)
        rc, text, got = emulate(z, len(raw))
        assert rc == 0 and got == raw, text
        m = re.search(r"chunks (\d+) \| tokens: (\d+) rounds", text)
        seen.append((int(m.group(1)), int(m.group(2))))
    # a constant frame is one 13-bit token (258 bytes, a fixed distance) over and over: a decoder started at a wrong bit never falls in step with it, the
    # chunk starts settle one per round -- 19 chunks, 19 rounds, the bound of stage A's loop (the stream is 6 KB); everything with content: 1-3 rounds
    assert max(c for c, _ in seen) > 500 and all(r <= 6 for c, r in seen if c > 100) and all(r <= c + 1 for c, r in seen), seen


def test_corrupt_and_foreign_streams(emulate):
    rng = np.random.default_rng(1)
    lits = [int(v) for v in rng.integers(0, 256, 4000)]
    ok = dt.zlib_stream(lits)
    assert emulate(ok, 4000)[0] == 0
    assert emulate(ok, 4004)[1].strip() == "status -4"                                         # another size than expected
    assert emulate(dt.zlib_stream(lits, end=False), 4000)[1].strip() == "status -3"            # no end-of-block code
    assert emulate(dt.zlib_stream(lits[:2000] + [("sym", 286)] + lits[2000:]), 4000)[1].strip() == "status -2"     # a length symbol that does not exist
    assert emulate(dt.zlib_stream(lits[:2000] + [("sym", 257), ("dist", 30)] + lits[2000:]), 4000)[1].strip() == "status -2"   # a distance code that does not exist
    assert emulate(dt.zlib_stream(lits[:10] + [(5, 11)] + lits[10:3995]), 4000)[1].strip() == "status -5"          # a match that reaches in front of the output
    assert emulate(zlib.compress(bytes(lits), 6), 4000)[0] == 2                                # a dynamic block: the host inflater's
    assert emulate(dt.zlib_stream(lits, header=(0, 1)), 4000)[0] == 2                          # not the final block
    junk = dt.zlib_stream(lits)[:-4] + bytes(rng.integers(0, 256, 600, dtype=np.uint8))        # whatever follows the end-of-block code is not decoded
    assert emulate(junk, 4000)[0] == 0


def test_mutated_streams_get_the_host_inflater_verdict(emulate):
    """Bit flips, byte pokes and truncations of a depth frame's stream: whatever the device's lane programs make of it is what the host inflater
    (sf_zlib_inflate, itself fuzzed against zlib: tools/fuzz_codecs.py) makes of it -- the same bytes, or both refuse.  A stream that stops being
    'one final fixed block' is the host's by construction (exit code 2)."""
    from scannet_amd import sens, synth
    from scannet_amd._abi import ScanfuseError
    raw = synth.render_room_depth(synth.trajectory_pose(11, 1200), 320, 240, noise_frame=5).tobytes()
    z = bytearray(sens.zlib_deflate(raw))
    rng = np.random.default_rng(42)
    agree_ok = agree_bad = foreign = 0
    for k in range(160):
        m = bytearray(z)
        kind = k % 4
        if kind == 0:
            i = int(rng.integers(2, len(m) - 4)); m[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            i = int(rng.integers(2, len(m) - 4)); m[i] = int(rng.integers(0, 256))
        elif kind == 2:
            m = m[: int(rng.integers(8, len(m)))]
        else:
            i = int(rng.integers(2, len(m) - 40)); m[i:i + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
        m = bytes(m)
        try:
            host = sens.zlib_inflate(m, len(raw))
            host_ok = len(host) == len(raw)
        except ScanfuseError:
            host, host_ok = None, False
        rc, text, got = emulate(m, len(raw))
        if rc == 2:
            foreign += 1
            continue
        assert (rc == 0) == host_ok, (k, kind, rc, text, host_ok)
        if rc == 0:
            assert got == host, (k, kind)
            agree_ok += 1
        else:
            agree_bad += 1
    assert agree_ok >= 5 and agree_bad >= 60, (agree_ok, agree_bad, foreign)
