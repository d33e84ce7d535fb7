import sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from scannet_amd import calibrate
import jpeg_tools
def cmp(name, blob, w, h):
    host = calibrate.jpeg_decode(blob, w, h)
    gpu = calibrate.jpeg_decode(blob, w, h, device=0)
    d = host != gpu
    print(name, "differing bytes", int(d.sum()), "of", d.size)
    if d.any():
        y, x, c = np.argwhere(d)[0]
        by, bx = (y // 8) * 8, (x // 8) * 8
        print(" first diff at", (int(y), int(x), int(c)), "block host:\n", host[by:by+8, bx:bx+8, c], "\n gpu:\n", gpu[by:by+8, bx:bx+8, c])
w, h = 16, 8
g = np.full((h, w), 77, np.uint8)
cmp("grey const", jpeg_tools.encode(g, ((1, 1),), qstep=1), w, h)
yy, xx = np.mgrid[0:h, 0:w]
cmp("grey ramp x", jpeg_tools.encode((xx * 9 + 20).astype(np.uint8), ((1, 1),), qstep=1), w, h)
cmp("grey ramp y", jpeg_tools.encode((yy * 20 + 20).astype(np.uint8), ((1, 1),), qstep=1), w, h)
rng = np.random.default_rng(0)
cmp("grey noise", jpeg_tools.encode(rng.integers(0, 256, (h, w), dtype=np.uint8), ((1, 1),), qstep=1), w, h)
cmp("grey noise q7", jpeg_tools.encode(rng.integers(0, 256, (h, w), dtype=np.uint8), ((1, 1),), qstep=7), w, h)
c = np.zeros((h, w, 3), np.uint8); c[...] = (200, 30, 60)
cmp("colour const 444", jpeg_tools.encode(c, ((1, 1), (1, 1), (1, 1)), qstep=1), w, h)
cmp("colour noise 444", jpeg_tools.encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), ((1, 1), (1, 1), (1, 1)), qstep=2), w, h)
cmp("colour noise 420", jpeg_tools.encode(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8), ((2, 2), (1, 1), (1, 1)), qstep=2), 16, 16)
