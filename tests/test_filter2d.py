"""2-D annotation filter: the kernels of AnnotationTools/Filter2dAnnotations/filter.cu that the tool calls.
CPU part: the checker oracle/filter2d_oracle.c (closed-form cases, its exp against libm, a golden digest).  GPU part (-m gpu):
scannet_amd/csrc/filter2d.hip against the checker, bit for bit.  PARITY UNPINNED against the reference binary (CUDA + mLib +
FreeImage; nvcc contraction and libdevice exp are not reproducible here)."""
import ctypes as C
import hashlib
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc
from scannet_amd import filter2d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "filter2d_golden.json")))
DW, DH, CW, CH = 80, 60, 162, 121     # small on purpose: the checker evaluates ~1e8 double exponentials per frame at this size


def _scene(seed=3):
    """A depth ramp with a step and holes, a two-tone colour image with noise, an instance image whose borders are ragged
    (what projecting a coarse annotated mesh produces: the filter's job is to snap them to depth / intensity edges)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:DH, 0:DW]
    depth = (1500 + 6 * xx + np.where(xx > DW // 2, 700, 0) + rng.integers(0, 4, (DH, DW))).astype(np.uint16)
    depth[rng.random((DH, DW)) < 0.01] = 0
    cy, cx = np.mgrid[0:CH, 0:CW]
    right = cx > CW // 2
    rgb = np.where(right[..., None], np.array([200, 60, 40]), np.array([40, 90, 200])).astype(np.int32) + rng.integers(-12, 13, (CH, CW, 3))
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    jitter = (6 * np.sin(cy / 5.0)).astype(int)
    inst = np.where(cx + jitter > CW // 2, 2, 1).astype(np.uint8)
    inst[cy < 8] = 0
    inst[(cy > 90) & (cx < 40)] = 5
    return depth, rgb, inst


def _tables():
    return filter2d.make_tables({0: 4, 1: 7, 4: 39})   # object ids 0, 1, 4 -> instances 1, 2, 5


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_exp_within_one_ulp_of_libm():
    L = orc.f2d_lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([-rng.random(20000) * 800, -rng.random(20000) * 30, -rng.random(20000), [0.0, -745.0, -746.0, -1e9, -1e-300]])
    worst = 0.0
    for x in xs:
        a, b = L.or_exp64(float(x)), math.exp(x) if x > -745.2 else 0.0
        if b == 0.0 or a == 0.0:
            assert abs(a - b) < 1e-320
            continue
        worst = max(worst, abs(a - b) / (np.nextafter(b, np.inf) - b))
        assert np.float32(a) == np.float32(b)
    assert worst <= 1.0
    assert math.isnan(L.or_exp64(float("nan"))) and L.or_exp64(float("-inf")) == 0.0


def test_kernels_known_answers():
    L = orc.f2d_lib()
    MINF = np.float32(-np.inf)
    # bilateral: a constant image is a fixed point; invalid pixels stay invalid and are ignored by their neighbours
    img = np.full((20, 30), 0.5, np.float32)
    img[5, 7] = MINF
    out = np.empty_like(img)
    L.or_f2d_bilateral(out.ctypes.data, img.ctypes.data, 2.0, 0.1, 30, 20)
    assert out[5, 7] == MINF and np.abs(out[np.isfinite(out)] - 0.5).max() < 1e-6
    # ... and an edge much larger than sigma_r survives (range weight exp(-1^2 / 0.02) = 2e-22)
    step = np.zeros((20, 30), np.float32)
    step[:, 15:] = 1.0
    L.or_f2d_bilateral(out.ctypes.data, step.ctypes.data, 2.0, 0.1, 30, 20)
    assert np.abs(out - step).max() < 1e-6
    # resample: identity at equal size; corners map to corners; nearest for labels
    ramp = np.add.outer(np.arange(12, dtype=np.float32), np.arange(16, dtype=np.float32))
    same = np.empty_like(ramp)
    L.or_f2d_resample_float(same.ctypes.data, 16, 12, ramp.ctypes.data, 16, 12)
    assert np.array_equal(same, ramp)
    up = np.zeros((23, 31), np.float32)
    L.or_f2d_resample_float(up.ctypes.data, 31, 23, ramp.ctypes.data, 16, 12)
    assert up[0, 0] == ramp[0, 0] and abs(up[-1, -1] - ramp[-1, -1]) < 1e-4 and abs(up[11, 15] - (11 * 11 / 22 + 15 * 15 / 30)) < 1e-3
    lab = (np.arange(12 * 16) % 7).astype(np.uint8).reshape(12, 16)
    upl = np.zeros((23, 31), np.uint8)
    L.or_f2d_resample_uchar(upl.ctypes.data, 31, 23, lab.ctypes.data, 16, 12)
    assert upl[0, 0] == lab[0, 0] and upl[22, 30] == lab[11, 15] and upl[10, 14] == lab[int(10 * 11 / 22 + 0.5), int(14 * 15 / 30 + 0.5)]
    # vote: a lone wrong pixel inside a uniform region is outvoted; across a depth edge nothing leaks
    to_idx, to_inst, to_label = _tables()
    inst = np.full((40, 40), 1, np.uint8)
    inst[:, 20:] = 2
    inst[10, 5] = 2           # speckle
    depth = np.where(np.arange(40)[None, :] < 20, 1.0, 2.0).astype(np.float32) * np.ones((40, 1), np.float32)
    inten = np.full((40, 40), 0.5, np.float32)
    out8 = np.empty_like(inst)
    L.or_f2d_vote(out8.ctypes.data, inst.ctypes.data, depth.ctypes.data, inten.ctypes.data, to_idx.ctypes.data, to_inst.ctypes.data, 6, 40, 40, 5.0, 0.1, 4.0)
    assert out8[10, 5] == 1 and (out8[:, :20] == 1).all() and (out8[:, 20:] == 2).all()
    lut = np.empty((40, 40), np.uint16)
    L.or_f2d_to_label(lut.ctypes.data, out8.ctypes.data, to_label.ctypes.data, 40, 40)
    assert (lut[:, :20] == 4).all() and (lut[:, 20:] == 7).all()


def test_frame_golden():
    depth, rgb, inst = _scene()
    io, lo = orc.f2d_frame(depth, rgb, inst, *_tables())
    assert _sha(io, lo) == GOLDEN["frame"]
    # the ragged instance border has been pulled onto the colour / depth edge at the image centre
    border_in = np.abs(np.argmax(inst[20:80] == 2, axis=1) - CW // 2)
    border_out = np.abs(np.argmax(io[20:80] == 2, axis=1) - CW // 2)
    assert border_out.mean() < border_in.mean() and border_out.max() <= 3
    assert set(np.unique(lo)) <= {0, 4, 7, 39}


def test_png_codec_against_pil(tmp_path):
    """FreeImageWrapper::loadImage / saveImage of the tool (Filter2dAnnotations.cpp:340-341,400-401): 8-bit instance, 16-bit label."""
    from PIL import Image
    rng = np.random.default_rng(0)
    a8 = (np.add.outer(np.arange(97), np.arange(131)) % 256).astype(np.uint8)
    a8[10:20, 30:60] = rng.integers(0, 256, (10, 30))
    a16 = (np.add.outer(np.arange(97), np.arange(131)) * 37 % 65536).astype(np.uint16)
    rgb = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    # files written by PIL (adaptive scan-line filters, several IDAT strategies) -> this reader
    for name, arr in (("a8", a8), ("a16", a16), ("rgb", rgb)):
        for level in (1, 9):
            p = str(tmp_path / ("%s_%d.png" % (name, level)))
            Image.fromarray(arr).save(p, compress_level=level)
            assert np.array_equal(filter2d.png_read(p), arr)
    # this writer -> PIL
    for name, arr in (("o8", a8), ("o16", a16), ("one", np.array([[7]], np.uint8))):
        p = str(tmp_path / (name + ".png"))
        filter2d.png_write_gray(p, arr)
        assert np.array_equal(np.array(Image.open(p)).astype(arr.dtype), arr)
        assert np.array_equal(filter2d.png_read(p), arr)
    # damaged files are refused
    blob = bytearray(open(str(tmp_path / "o8.png"), "rb").read())
    blob[40] ^= 0xFF
    open(str(tmp_path / "bad.png"), "wb").write(bytes(blob))
    with pytest.raises(Exception):
        filter2d.png_read(str(tmp_path / "bad.png"))
    open(str(tmp_path / "not.png"), "wb").write(b"hello world, this is not a png file at all....")
    with pytest.raises(Exception, match="not a PNG"):
        filter2d.png_read(str(tmp_path / "not.png"))


@pytest.mark.gpu
def test_tool_end_to_end(tmp_path):
    """bin/filter2dannotations on a three-frame scene: .sens + projected annotation PNGs + aggregation JSON + label map in, filtered
    PNGs out, every frame compared with the checker; an invalid pose gives empty images; a finished scene is skipped."""
    import subprocess
    from scannet_amd import sens
    K = np.eye(4, dtype=np.float32)
    sd = sens.SensorData.create(CW, CH, DW, DH, K, K, color_compression=0, depth_compression=1)
    scenes = [_scene(seed=s) for s in (3, 4, 5)]
    for i, (d, rgb, inst) in enumerate(scenes):
        pose = np.eye(4, dtype=np.float32) if i != 1 else np.full((4, 4), -np.inf, np.float32)
        sd.add_frame(d, pose, color=rgb)
    sens_path = str(tmp_path / "scene0.sens")
    sd.save(sens_path)
    sd.close()
    ann = tmp_path / "annotations-2d" / "scene0"
    (ann / "instance").mkdir(parents=True)
    (ann / "label").mkdir()
    for i, (_, _, inst) in enumerate(scenes):
        filter2d.png_write_gray(str(ann / "instance" / ("%d.png" % i)), inst)
        filter2d.png_write_gray(str(ann / "label" / ("%d.png" % i)), inst.astype(np.uint16) * 3)
    agg = {"sceneId": "scannet.scene0", "appId": "Aggregator.v2", "segGroups": [
        {"id": 0, "objectId": 0, "segments": [1, 2, 3], "label": "chair"}, {"id": 1, "objectId": 1, "segments": [4], "label": "the \"big\" table"},
        {"id": 4, "objectId": 4, "segments": [], "label": "unknown thing"}], "segmentsFile": "scannet.scene0_vh_clean_2.0.010000.segs.json"}
    (tmp_path / "scene0.aggregation.json").write_text(json.dumps(agg))
    # label map: id = 1-based line number of the `category` column (LabelUtil.h:40-84)
    rows = ["id\tcategory\tcount", "x\twall\t9", "x\tchair\t8", "x\t\t0", "x\tthe \"big\" table\t7"]
    (tmp_path / "labels.tsv").write_text("\n".join(rows) + "\n")
    out = tmp_path / "annotations-2d-filtered" / "scene0"
    (tmp_path / "annotations-2d-filtered").mkdir()
    exe = os.path.join(ROOT, "bin", "filter2dannotations")
    cmd = [exe, str(ann), sens_path, str(tmp_path / "scene0.aggregation.json"), str(tmp_path / "labels.tsv"), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    assert "read 3 labels" in r.stdout and "3 frames" in r.stdout
    tables = filter2d.make_tables({0: 2, 1: 4, 4: 0})       # chair = line 2, the "big" table = line 4, unknown thing -> 0
    for i, (d, rgb, inst) in enumerate(scenes):
        gi = filter2d.png_read(str(out / "instance" / ("%d.png" % i)))
        gl = filter2d.png_read(str(out / "label" / ("%d.png" % i)))
        if i == 1:
            assert not gi.any() and not gl.any()
            continue
        oi, ol = orc.f2d_frame(d, rgb, inst, *tables)
        assert np.array_equal(gi, oi) and np.array_equal(gl, ol), i
    again = subprocess.run(cmd, capture_output=True, text=True)
    assert again.returncode == 0 and "skipping, already exists" in again.stdout
    bad = subprocess.run([exe, str(tmp_path / "nope"), sens_path, "a", "b", str(out)], capture_output=True, text=True)
    assert bad.returncode != 0 and "instance/label dir does not exist" in bad.stderr


@pytest.mark.gpu
def test_gpu_matches_the_checker_bit_for_bit():
    depth, rgb, inst = _scene()
    tables = _tables()
    with filter2d.Filter2d((DW, DH), (CW, CH)) as f:
        f.set_tables(*tables)
        io, lo, us = f.frame(depth, rgb, inst)
        oi, ol = orc.f2d_frame(depth, rgb, inst, *tables)
        assert np.array_equal(io, oi), "%d instance pixels differ" % (io != oi).sum()
        assert np.array_equal(lo, ol)
        assert _sha(io, lo) == GOLDEN["frame"]
        d2, r2, i2 = _scene(seed=11)           # a second frame through the same (reused) buffers
        io2, lo2, _ = f.frame(d2, r2, i2)
        oi2, ol2 = orc.f2d_frame(d2, r2, i2, *tables)
        assert np.array_equal(io2, oi2) and np.array_equal(lo2, ol2)
