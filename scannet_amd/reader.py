"""`python -m scannet_amd.reader` -- the command line of SensReader/python/reader.py (:7-40) over this library, in Python 3:

    python -m scannet_amd.reader --filename scene0000_00.sens --output_path out [--export_depth_images] [--export_color_images] [--export_poses] [--export_intrinsics]

writes out/depth/<i>.png (16-bit), out/color/<i>.jpg, out/pose/<i>.txt, out/intrinsic/{intrinsic,extrinsic}_{color,depth}.txt, as the reference does
(SensorData.py:78-124).  `bin/sens <file> <outDir>` is the C++ exporter's drop-in (frame-%06d.color.jpg / .depth.pgm / .pose.txt + _info.txt)."""
import argparse
import os
import sys

from .sens import SensorData


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--filename', required=True, help='path to sens file to read')
    parser.add_argument('--output_path', required=True, help='path to output folder')
    parser.add_argument('--export_depth_images', dest='export_depth_images', action='store_true')
    parser.add_argument('--export_color_images', dest='export_color_images', action='store_true')
    parser.add_argument('--export_poses', dest='export_poses', action='store_true')
    parser.add_argument('--export_intrinsics', dest='export_intrinsics', action='store_true')
    parser.set_defaults(export_depth_images=False, export_color_images=False, export_poses=False, export_intrinsics=False)
    opt = parser.parse_args(argv)
    print(opt)
    os.makedirs(opt.output_path, exist_ok=True)
    sys.stdout.write('loading %s...' % opt.filename)
    sd = SensorData(opt.filename)
    sys.stdout.write('loaded!\n')
    n = len(sd.frames)
    if opt.export_depth_images:
        print('exporting', n, ' depth frames to', os.path.join(opt.output_path, 'depth'))
        sd.export_depth_images(os.path.join(opt.output_path, 'depth'))
    if opt.export_color_images:
        print('exporting', n, 'color frames to', os.path.join(opt.output_path, 'color'))
        sd.export_color_images(os.path.join(opt.output_path, 'color'))
    if opt.export_poses:
        print('exporting', n, 'camera poses to', os.path.join(opt.output_path, 'pose'))
        sd.export_poses(os.path.join(opt.output_path, 'pose'))
    if opt.export_intrinsics:
        print('exporting camera intrinsics to', os.path.join(opt.output_path, 'intrinsic'))
        sd.export_intrinsics(os.path.join(opt.output_path, 'intrinsic'))
    sd.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
