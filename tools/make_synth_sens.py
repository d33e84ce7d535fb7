#!/usr/bin/env python3
"""Write a synthetic .sens stream with this repo's writer.

  python tools/make_synth_sens.py out.sens --config plane --frames 100          # config 1: 2 m plane, identity poses
  python tools/make_synth_sens.py out.sens --config room --frames 200 [--total 5578] [--noise] [--size 640 480]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scannet_amd import sens, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--config", choices=["plane", "room"], default="room")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--total", type=int, default=0, help="length of the full walk the frames are taken from (default: --frames)")
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--size", type=int, nargs=2, default=[640, 480])
    ap.add_argument("--invalid-every", type=int, default=0, help="mark every n-th pose as tracking-lost (-inf)")
    a = ap.parse_args()
    W, H = a.size
    K = synth.intrinsic_matrix(W, H)
    sd = sens.SensorData.create(0, 0, W, H, K, K, color_compression=0, depth_compression=1, sensor_name="StructureSensor")
    total = a.total or a.frames
    for i in range(a.frames):
        if a.config == "plane":
            d, pose = synth.plane_frame(W, H), np.eye(4, dtype=np.float32)
        else:
            pose = synth.trajectory_pose(i, total)
            d = synth.render_room_depth(pose, W, H, noise_frame=i if a.noise else None)
        if a.invalid_every and i % a.invalid_every == a.invalid_every - 1:
            pose = np.full((4, 4), -np.inf, np.float32)
        sd.add_frame(d, pose, timestamp_color=33333 * i, timestamp_depth=33333 * i)
    sd.save(a.out)
    print("wrote %s: %d frames %dx%d" % (a.out, a.frames, W, H))


if __name__ == "__main__":
    main()
