"""End-to-end on the GPU: .sens -> threaded decode -> fusion -> mesh -> PLY -> Segmentator, through the library
entry point and through the two drop-in executables (the `improve` and `segment` stages of Server/scan_processor.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

from scannet_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_sens(path, n, W, H, total, invalid=()):
    from scannet_amd import sens
    K = synth.intrinsic_matrix(W, H)
    sd = sens.SensorData.create(0, 0, W, H, K, K, sensor_name="StructureSensor")
    frames = []
    for i in range(n):
        pose = synth.trajectory_pose(i * 7, total)
        d = synth.render_room_depth(pose, W, H, noise_frame=i)
        if i in invalid:
            pose = np.full((4, 4), -np.inf, np.float32)
        sd.add_frame(d, pose, timestamp_depth=i)
        frames.append((d, pose))
    sd.save(path)
    return frames


def test_fuse_run_matches_frame_by_frame(oracle, tmp_path):
    from scannet_amd import fusion, sens
    W, H = 320, 240
    p = str(tmp_path / "scene.sens")
    frames = _write_sens(p, 40, W, H, 1200, invalid=(3, 17))
    fx, fy, mx, my = synth.intrinsics(W, H)
    gp = fusion.default_params(depth_width=W, depth_height=H, fx=fx, fy=fy, mx=mx, my=my, voxel_size=0.008, num_sdf_blocks=1 << 17)
    sd = sens.SensorData(p)
    with fusion.Fuser(gp) as a, fusion.Fuser(gp) as b:
        st = a.run(sd, decode_threads=5)
        assert (st["frames_total"], st["frames_integrated"], st["frames_skipped"]) == (40, 38, 2)
        for d, pose in frames:
            b.integrate(d, pose)
        ca, va = a.export_blocks()
        cb, vb = b.export_blocks()
        assert np.array_equal(ca, cb) and np.array_equal(va.view(np.uint8), vb.view(np.uint8))
        assert a.stats()["frames_skipped"] == 2
        # sub-range + wrong resolution
        with fusion.Fuser(gp) as c:
            assert c.run(sd, first=10, last=20, decode_threads=2)["frames_total"] == 10
        gp2 = fusion.default_params(depth_width=W + 8, depth_height=H, fx=fx, fy=fy, mx=mx, my=my, num_sdf_blocks=1 << 12)
        with fusion.Fuser(gp2) as d2:
            with pytest.raises(Exception, match="created for"):
                d2.run(sd)


def test_drop_in_executables(tmp_path):
    """`depthsensing p1 p2 scan.sens` -> scan_vh.ply, then `segmentator scan_vh.ply` -> scan_vh.0.010000.segs.json;
    stderr stays empty (Server/util.py:42-44 logs stderr as an error)."""
    W, H = 320, 240
    sens_path = str(tmp_path / "scene0000_00.sens")
    _write_sens(sens_path, 30, W, H, 1200)
    params = tmp_path / "zParametersScanNet.txt"
    ref_params = "/root/reference/Server/tools/recons/zParametersScanNet.txt"
    text = open(ref_params).read() if os.path.exists(ref_params) else "s_SDFVoxelSize = 0.010f;\ns_SDFTruncation = 0.06f;\ns_SDFTruncationScale = 0.02f;\n"
    params.write_text(text + "\ns_hashNumSDFBlocks = 200000;\n")
    (tmp_path / "zParametersTrackingDefault.txt").write_text("// tracking parameters are not used by the fusion stage\n")
    out = subprocess.run([os.path.join(ROOT, "bin", "depthsensing"), str(params), str(tmp_path / "zParametersTrackingDefault.txt"), sens_path],
                         capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    assert out.stderr == ""
    ply = str(tmp_path / "scene0000_00_vh.ply")
    assert os.path.getsize(ply) > 100000
    assert "Integrated 30 frames" in out.stdout and "written to" in out.stdout
    # meshlabserver -i X_vh.ply -o X_vh_clean.ply -m vc -s clean.mlx (scan_processor.py:143)
    mlx = tmp_path / "clean.mlx"
    mlx.write_text('<!DOCTYPE FilterScript>\n<FilterScript>\n <filter name="Merge Close Vertices">\n  <Param name="Threshold" value="0.0010689" type="RichAbsPerc"/>\n </filter>\n'
                   ' <filter name="Remove Duplicate Faces"/>\n <filter name="Remove Isolated pieces (wrt Face Num.)">\n  <Param name="MinComponentSize" value="7500" type="RichInt"/>\n </filter>\n'
                   ' <filter name="Remove Unreferenced Vertex"/>\n</FilterScript>\n')
    clean_ply = str(tmp_path / "scene0000_00_vh_clean.ply")
    cl = subprocess.run([os.path.join(ROOT, "bin", "meshclean"), "-i", ply, "-o", clean_ply, "-m", "vc", "-s", str(mlx)], capture_output=True, text=True)
    assert cl.returncode == 0 and cl.stderr == "", cl.stderr
    from scannet_amd import segmentator
    nv0, nf0 = segmentator.Mesh.read(ply).counts()
    nv, nf = segmentator.Mesh.read(clean_ply).counts()
    assert 10000 < nf <= nf0 and nv <= nv0
    seg = subprocess.run([os.path.join(ROOT, "bin", "segmentator"), clean_ply], capture_output=True, text=True)
    assert seg.returncode == 0 and seg.stderr == ""
    js = json.load(open(str(tmp_path / "scene0000_00_vh_clean.0.010000.segs.json")))
    assert js["params"] == {"kThresh": 0.01, "segMinVerts": 20} and js["sceneId"] == "/scene0000_00_vh_clean"
    assert len(js["segIndices"]) == nv
    # failure protocol: non-zero exit and a message on stderr
    bad = subprocess.run([os.path.join(ROOT, "bin", "depthsensing"), str(params), str(params), str(tmp_path / "missing.sens")], capture_output=True, text=True)
    assert bad.returncode != 0 and "could not open" in bad.stderr
