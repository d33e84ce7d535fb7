#!/usr/bin/env python3
"""Mutation fuzzing of the parsers a .sens file (and the mesh tools) feed with untrusted bytes -- baseline JPEG, zlib inflate, PNG, PLY / OBJ (and the Segmentator
behind them), the Occipital depth code, the .sens container, ScannerApp captures, parameter files and filter scripts -- against the ASan + UBSan build of the host code (tools/sanitize.py): valid inputs are truncated, bit-flipped,
spliced and length-poked; every call must return (SF_OK or an error), never trip a sanitizer.

    python tools/fuzz_codecs.py [iterations per codec = 4000] [seed = 1]      # builds the sanitizer library if needed, re-executes under libasan
"""
import ctypes as C
import io
import os
import subprocess
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _reexec():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sanitize
    lib, _ = sanitize.build()
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, SCANFUSE_LIBRARY=lib, SF_FUZZ_CHILD="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode


def mutate(rng, blob):
    b = bytearray(blob)
    k = rng.integers(0, 6)
    if k == 0 and len(b) > 4:
        del b[int(rng.integers(1, len(b))):]                                  # truncate
    elif k == 1:
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))      # bit flips
    elif k == 2:
        i = int(rng.integers(0, len(b)))
        b[i:i + 2] = bytes(rng.integers(0, 256, 2, dtype="uint8"))              # poke two bytes (lengths, markers)
    elif k == 3 and len(b) > 16:
        i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
        b[i:j] = b[j:j + (j - i)]                                                 # splice
    elif k == 4:
        i = int(rng.integers(0, len(b)))
        b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype="uint8"))   # insert noise
    else:
        i = int(rng.integers(0, len(b)))
        b[i:i + 4] = b"\xff\xff\xff\xff"                                   # saturate a field
    return bytes(b)


def fix_png_crcs(blob):
    """Recompute the CRC of every chunk whose framing survived the mutation, so that the mutated bytes reach the inflater and the filters."""
    import struct
    b = bytearray(blob)
    pos = 8
    while pos + 12 <= len(b):
        (n,) = struct.unpack(">I", b[pos:pos + 4])
        if pos + 12 + n > len(b):
            break
        b[pos + 8 + n:pos + 12 + n] = struct.pack(">I", zlib.crc32(bytes(b[pos + 4:pos + 8 + n])) & 0xFFFFFFFF)
        pos += 12 + n
    return bytes(b)


def main():
    import numpy as np
    from PIL import Image
    from scannet_amd import _abi
    from tests import jpeg_tools
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    L = _abi.lib()
    u8p, vp = C.POINTER(C.c_uint8), C.c_void_p
    L.sf_jpeg_decode.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    L.sf_zlib_inflate.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sf_png_read.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp)]
    L.sf_free.argtypes = [vp]
    L.sf_ply_read.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.sf_mesh_free.argtypes = [vp]
    L.sf_occ_decode.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
    tmp = tempfile.mkdtemp(prefix="sf_fuzz_")
    counts = {}

    def tally(name, rc):
        c = counts.setdefault(name, [0, 0])
        c[0 if rc == 0 else 1] += 1

    # ---- JPEG: PIL baseline files of several layouts + the test encoder's unusual ones
    W, H = 48, 40
    img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    seeds = []
    for sub, q in ((0, 90), (1, 75), (2, 50)):
        bio = io.BytesIO()
        Image.fromarray(img).save(bio, "JPEG", quality=q, subsampling=sub)
        seeds.append(bio.getvalue())
    bio = io.BytesIO()
    Image.fromarray(img[:, :, 0]).save(bio, "JPEG", quality=80)
    seeds.append(bio.getvalue())
    seeds.append(jpeg_tools.encode(img, ((2, 1), (1, 1), (1, 1)), qstep=3, restart=2))
    out = np.zeros(W * H * 3 + 64, np.uint8)
    for it in range(n_iter):
        blob = mutate(rng, seeds[it % len(seeds)])
        buf = np.frombuffer(blob, np.uint8).copy()          # an exact-size heap block: an over-read is an ASan report
        tally("jpeg", L.sf_jpeg_decode(buf.ctypes.data, len(buf), W, H, out.ctypes.data))
    # ---- zlib
    raw = (rng.integers(0, 40, 20000)).astype(np.uint16).tobytes()
    zs = [zlib.compress(raw, lvl) for lvl in (0, 1, 6, 9)]
    dst = np.zeros(len(raw) + 16, np.uint8)
    nout = C.c_uint64(0)
    for it in range(n_iter):
        blob = mutate(rng, zs[it % len(zs)])
        buf = np.frombuffer(blob, np.uint8).copy()
        tally("inflate", L.sf_zlib_inflate(buf.ctypes.data, len(buf), dst.ctypes.data, int(rng.choice([len(raw), len(raw) // 2, 16])), C.byref(nout)))
    # ---- PNG (file based)
    pngs = []
    for mode, arr in (("RGB", img), ("L", img[:, :, 0]), ("I;16", (img[:, :, 0].astype(np.uint16) * 257)), ("P", img[:, :, 0] % 16)):
        bio = io.BytesIO()
        im = Image.fromarray(arr)
        if mode == "P":
            im = im.convert("P")
        im.save(bio, "PNG", interlace=(mode == "RGB"))
        pngs.append(bio.getvalue())
    w, h, ch, bits, data = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int(), vp()
    path = os.path.join(tmp, "f.png").encode()
    for it in range(n_iter // 2):
        blob = mutate(rng, pngs[it % len(pngs)])
        open(path, "wb").write(fix_png_crcs(blob) if it % 4 else blob)
        rc = L.sf_png_read(path, C.byref(w), C.byref(h), C.byref(ch), C.byref(bits), C.byref(data))
        if rc == 0:
            L.sf_free(data)
        tally("png", rc)
    # ---- PLY (ascii and binary)
    from tests import meshes
    v, t = meshes.grid(6)
    plys = []
    for fmt in ("ascii", "le", "be"):
        p = os.path.join(tmp, "m_%s.ply" % fmt)
        meshes.write_ply(p, v, t, fmt=fmt, colors=True)
        plys.append(open(p, "rb").read())
    path = os.path.join(tmp, "f.ply").encode()
    mh = vp()
    for it in range(n_iter // 2):
        open(path, "wb").write(mutate(rng, plys[it % len(plys)]))
        rc = L.sf_ply_read(path, C.byref(mh))
        if rc == 0:
            L.sf_mesh_free(mh)
        tally("ply", rc)
    # ---- Occipital depth code
    depth = (1000 + (np.arange(64 * 48) % 97)).astype(np.uint16)
    L.sf_occ_encode_bound.argtypes = [C.c_uint64]
    L.sf_occ_encode_bound.restype = C.c_uint64
    L.sf_occ_encode.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    enc = np.zeros(int(L.sf_occ_encode_bound(depth.size)), np.uint8)
    used = C.c_uint64(0)
    assert L.sf_occ_encode(depth.ctypes.data, depth.size, enc.ctypes.data, enc.size, C.byref(used)) == 0
    code = enc[: used.value].tobytes()
    dout = np.zeros(depth.size, np.uint16)
    for it in range(n_iter):
        blob = mutate(rng, code)
        buf = np.frombuffer(blob, np.uint8).copy()
        tally("occipital", L.sf_occ_decode(buf.ctypes.data, len(buf), dout.size, dout.ctypes.data))
    # ---- the .sens container itself (header, frame table, per-frame blob sizes), every frame of whatever opens decoded
    from scannet_amd import sens, synth
    Ws, Hs = 32, 24
    K = synth.intrinsic_matrix(Ws, Hs)
    sd = sens.SensorData.create(Ws, Hs, Ws, Hs, K, K, sensor_name="fuzz", color_compression=0)
    for i in range(5):
        sd.add_frame((1000 + (np.arange(Ws * Hs) * (i + 1)) % 500).astype(np.uint16).reshape(Hs, Ws), np.eye(4, dtype=np.float32), timestamp_depth=i,
                     color=(rng.random((Hs, Ws, 3)) * 255).astype(np.uint8))
    good = os.path.join(tmp, "good.sens")
    sd.save(good)
    blob0 = open(good, "rb").read()
    L.sf_sens_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.sf_sens_close.argtypes = [vp]
    L.sf_sens_get_info.argtypes = [vp, C.POINTER(sens.SfSensInfo)]
    L.sf_sens_decode_depth.argtypes = [vp, C.c_uint64, vp]
    L.sf_sens_decode_color.argtypes = [vp, C.c_uint64, vp]
    path = os.path.join(tmp, "f.sens").encode()
    for it in range(n_iter // 2):
        open(path, "wb").write(mutate(rng, blob0))
        h = vp()
        rc = L.sf_sens_open(path, C.byref(h))
        if rc == 0:
            info = sens.SfSensInfo()
            L.sf_sens_get_info(h, C.byref(info))
            npx_d, npx_c = int(info.depth_width) * int(info.depth_height), int(info.color_width) * int(info.color_height)
            if 0 < npx_d <= 1 << 22 and npx_c <= 1 << 22:
                dd, cc = np.zeros(npx_d, np.uint16), np.zeros(max(npx_c, 1) * 3, np.uint8)
                for fr in range(min(int(info.num_frames), 8)):
                    L.sf_sens_decode_depth(h, fr, dd.ctypes.data)
                    L.sf_sens_decode_color(h, fr, cc.ctypes.data)
            L.sf_sens_close(h)
        tally("sens", rc)
    # ---- text parsers: parameter files and MeshLab filter scripts
    from scannet_amd import _abi as abi2, meshclean
    ptxt = b"s_SDFVoxelSize = 0.004f;\ns_hashNumBuckets = 524288;\ns_integrationWidth = 640;\ns_SDFTruncation = 0.06f;\n// c\ns_sensorIdx = 8;\n"
    mtxt = (b'<!DOCTYPE FilterScript>\n<FilterScript>\n <filter name="Merge Close Vertices">\n  <Param name="Threshold" value="0.0010689" type="RichAbsPerc"/>\n </filter>\n'
            b' <filter name="Remove Duplicate Faces"/>\n <filter name="Remove Isolated pieces (wrt Face Num.)">\n  <Param name="MinComponentSize" value="7500"/>\n </filter>\n'
            b' <filter name="Remove Unreferenced Vertex"/>\n</FilterScript>\n')
    L.sf_params_load_file.argtypes = [C.c_char_p, vp]
    L.sf_mlx_load.argtypes = [C.c_char_p, vp]
    pp, ms = abi2.SfParams(), meshclean.SfCleanScript()
    path = os.path.join(tmp, "f.txt").encode()
    for it in range(n_iter // 2):
        open(path, "wb").write(mutate(rng, ptxt if it & 1 else mtxt))
        tally("text", L.sf_params_load_file(path, C.byref(pp)) if it & 1 else L.sf_mlx_load(path, C.byref(ms)))
    # ---- a ScannerApp capture (<base>.txt + .depth): the metadata file or the frame stream damaged, every frame of whatever opens decoded
    # (round 5: numDepthFrames = 0 reached memcpy with a NULL destination in sf_capture_open)
    from scannet_amd import capture
    base = os.path.join(tmp, "cap")
    meta = [tuple(x.split(" = ")) for x in ("colorWidth = 32", "colorHeight = 24", "depthWidth = 32", "depthHeight = 24", "fx_color = 30", "fy_color = 30", "mx_color = 16",
                                             "my_color = 12", "fx_depth = 30", "fy_depth = 30", "mx_depth = 16", "my_depth = 12", "numDepthFrames = 2", "numColorFrames = 2",
                                             "numIMUmeasurements = 0")]
    capture.write_capture(base, [rng.integers(0, 2048, (24, 32)).astype(np.uint16) for _ in range(2)], [0.1, 0.2], meta)
    good = {ext: open(base + ext, "rb").read() for ext in (".txt", ".depth")}
    cl = capture._lib()
    for it in range(n_iter // 4):
        ext = ".txt" if it & 1 else ".depth"
        open(base + ext, "wb").write(mutate(rng, good[ext]))
        h = vp()
        rc = cl.sf_capture_open((base + ".txt").encode(), C.byref(h))
        if rc == 0:
            m = capture.SfCaptureMeta()
            cl.sf_capture_get_meta(h, C.byref(m))
            npx = int(m.depth_width) * int(m.depth_height)
            if 0 < npx <= 1 << 22:
                dd, ts = np.zeros(npx, np.uint16), C.c_uint64(0)
                for fr in range(min(int(m.num_depth_frames), 4)):
                    cl.sf_capture_decode_depth(h, fr, dd.ctypes.data, C.byref(ts))
            cl.sf_capture_close(h)
        open(base + ext, "wb").write(good[ext])
        tally("capture", rc)
    # ---- .obj through the mesh reader, and whatever mesh comes out through the Segmentator (indices of a damaged file must not reach it unchecked)
    obj = ("".join("v %f %f %f\n" % tuple(x) for x in v) + "".join("f %d %d %d\n" % tuple(int(i) + 1 for i in f) for f in t)).encode()
    L.sf_mesh_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sf_mesh_copy.argtypes = [vp, vp, vp, vp, vp]
    L.sf_segment_mesh.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_float, C.c_int, vp]
    path = os.path.join(tmp, "f.obj").encode()
    path_ply = os.path.join(tmp, "g.ply").encode()
    for it in range(n_iter // 4):
        src, pth = (obj, path) if it & 1 else (plys[0], path_ply)
        open(pth, "wb").write(mutate(rng, src))
        rc = L.sf_ply_read(pth, C.byref(mh))
        if rc == 0:
            nv, nf = C.c_uint64(0), C.c_uint64(0)
            L.sf_mesh_counts(mh, C.byref(nv), C.byref(nf))
            if nv.value < 1 << 20 and nf.value < 1 << 20:
                xyz, tri, seg = np.zeros(max(nv.value, 1) * 3, np.float32), np.zeros(max(nf.value, 1) * 3, np.uint32), np.zeros(max(nv.value, 1), np.int32)
                L.sf_mesh_copy(mh, xyz.ctypes.data, None, tri.ctypes.data, None)
                L.sf_segment_mesh(xyz.ctypes.data, nv.value, tri.ctypes.data, nf.value, 0.01, 20, seg.ctypes.data)
            L.sf_mesh_free(mh)
        tally("obj/segment", rc)
    # ---- a folder of images -> .sens (sf_sens_load_from_images): info.txt, a pose file or a depth PGM damaged
    from scannet_amd import calibrate
    folder = os.path.join(tmp, "images")
    os.makedirs(folder, exist_ok=True)
    L.sf_sens_load_from_images.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    K4 = synth.intrinsic_matrix(16, 12)
    mat = lambda m: " ".join("%g" % x for x in np.asarray(m).reshape(-1))
    files = {"info.txt": ("m_versionNumber = 4\nm_sensorName = fuzz\nm_colorWidth = 16\nm_colorHeight = 12\nm_depthWidth = 16\nm_depthHeight = 12\nm_depthShift = 1000\n"
                          "m_calibrationColorIntrinsic = %s \nm_calibrationColorExtrinsic = %s \nm_calibrationDepthIntrinsic = %s \nm_calibrationDepthExtrinsic = %s \n"
                          "m_frames.size = 2\n" % (mat(K4), mat(np.eye(4)), mat(K4), mat(np.eye(4)))).encode()}
    for i in range(2):
        files["frame-%06d.color.jpg" % i] = calibrate.jpeg_encode((rng.random((12, 16, 3)) * 255).astype(np.uint8), 80, True)
        files["frame-%06d.depth.pgm" % i] = b"P5\n# c\n16 12\n65535\n" + (rng.integers(0, 65536, 192)).astype(">u2").tobytes()
        files["frame-%06d.pose.txt" % i] = b"1 0 0 0.5\n0 1 0 -inf\n0 0 1 2\n0 0 0 1"
    for name, body in files.items():
        open(os.path.join(folder, name), "wb").write(body)
    names = [n for n in files if not n.endswith(".jpg")]
    for it in range(n_iter // 4):
        name = names[it % len(names)]
        open(os.path.join(folder, name), "wb").write(mutate(rng, files[name]))
        h = vp()
        rc = L.sf_sens_load_from_images(folder.encode(), None, None, C.byref(h))
        if rc == 0:
            L.sf_sens_close(h)
        open(os.path.join(folder, name), "wb").write(files[name])
        tally("images", rc)
    for name, (ok, err) in counts.items():
        print("fuzz %-10s %6d decoded, %6d rejected" % (name, ok, err))
    print("fuzz: no sanitizer report")
    return 0


if __name__ == "__main__":
    sys.exit(main() if os.environ.get("SF_FUZZ_CHILD") else _reexec())
