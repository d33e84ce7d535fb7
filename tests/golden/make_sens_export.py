#!/usr/bin/env python3
"""Makes tests/golden/sens_export/: a small .sens (14 frames of 32x24 JPEG colour + zlib depth, one frame with the all -inf "tracking lost" pose, a sensor
name with a blank) and what the REFERENCE exporter writes for it -- SensReader/c++ `sens <file> <outDir>`, compiled from /root/reference by
oracle/Makefile into oracle/_ref/sens_ref -- so that tests/test_sens_export.py can hold bin/sens against the reference's bytes where the reference
is not there (the GPU box).  Run in the build container:   python tests/golden/make_sens_export.py"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scannet_amd import calibrate, sens, synth   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sens_export")


def write_scan(path, n=14, W=32, H=24):
    rng = np.random.default_rng(2026)
    K = synth.intrinsic_matrix(W, H)
    pictures = [calibrate.jpeg_encode(rng.integers(0, 256, (H, W, 3), dtype=np.uint8), 90, True) for _ in range(3)]
    sd = sens.SensorData.create(W, H, W, H, K, K, color_compression=2, depth_compression=1, sensor_name="Structure Sensor")
    for i in range(n):
        pose = synth.trajectory_pose(i * 61, 1200) if i != 4 else np.full((4, 4), -np.inf, np.float32)
        sd.add_frame(rng.integers(0, 6000, (H, W), dtype=np.uint16), pose, color=pictures[i % 3], timestamp_color=i * 33333 + 7, timestamp_depth=i * 33333)
    sd.save(path)
    sd.close()


def main():
    ref = os.path.join(ROOT, "oracle", "_ref", "sens_ref")
    if not os.path.exists(ref):
        raise SystemExit("oracle/_ref/sens_ref is not built (make -C oracle ref, needs /root/reference)")
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    scan = os.path.join(OUT, "scan.sens")
    write_scan(scan)
    r = subprocess.run([ref, "scan.sens", "reference_out"], capture_output=True, cwd=OUT, check=True)
    open(os.path.join(OUT, "reference_stdout.txt"), "wb").write(r.stdout)
    print("wrote", OUT, len(os.listdir(os.path.join(OUT, "reference_out"))), "files")


if __name__ == "__main__":
    main()
