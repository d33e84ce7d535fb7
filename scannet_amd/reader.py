"""`python -m scannet_amd.reader` -- the command line of SensReader/python/reader.py (:7-40) over this library, in Python 3:

    python -m scannet_amd.reader --filename scene0000_00.sens --output_path out [--export_depth_images] [--export_color_images] [--export_poses] [--export_intrinsics]

writes out/depth/<i>.png (16-bit), out/color/<i>.jpg, out/pose/<i>.txt, out/intrinsic/{intrinsic,extrinsic}_{color,depth}.txt, as the reference does
(SensorData.py:78-124).  `bin/sens <file> <outDir>` is the C++ exporter's drop-in (frame-%06d.color.jpg / .depth.pgm / .pose.txt + _info.txt)."""
import argparse
import os
import sys

from .sens import SensorData

# switch of the reference's command line -> (sub-folder of --output_path, SensorData method, what the progress line calls the items)
EXPORTS = (
    ("export_depth_images", "depth", "export_depth_images", " depth frames"),
    ("export_color_images", "color", "export_color_images", "color frames"),
    ("export_poses", "pose", "export_poses", "camera poses"),
    ("export_intrinsics", "intrinsic", "export_intrinsics", None),
)


def parse(argv):
    ap = argparse.ArgumentParser(prog="python -m scannet_amd.reader", description="export the contents of a .sens file")
    ap.add_argument("--filename", required=True, help="path to sens file to read")
    ap.add_argument("--output_path", required=True, help="path to output folder")
    for switch, folder, _, _ in EXPORTS:
        ap.add_argument("--" + switch, action="store_true", default=False, help="write <output_path>/%s" % folder)
    return ap.parse_args(argv)


def main(argv=None):
    opt = parse(argv)
    print(opt)
    os.makedirs(opt.output_path, exist_ok=True)
    sys.stdout.write("loading %s..." % opt.filename)
    with_frames = SensorData(opt.filename)
    sys.stdout.write("loaded!\n")
    try:
        for switch, folder, method, items in EXPORTS:
            if not getattr(opt, switch):
                continue
            target = os.path.join(opt.output_path, folder)
            if items is None:
                print("exporting camera intrinsics to", target)
            else:
                print("exporting", len(with_frames.frames), items, "to", target)
            getattr(with_frames, method)(target)
    finally:
        with_frames.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
