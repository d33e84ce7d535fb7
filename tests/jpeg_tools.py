"""A small baseline JPEG writer for the tests: arbitrary sampling factors per component (4:4:0, 4:1:1, 4:1:0, luma-subsampled ...),
grey pictures, restart intervals -- layouts PIL cannot produce, which the reference decoder (stb_image) sends down its less common
resampling paths.  Huffman tables are the simplest valid ones (every DC category a 4-bit code, every AC symbol an 8-bit code), so
the streams are large but any conforming decoder reads them.  Test infrastructure only."""
import numpy as np
from scipy.fft import dctn

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
AC_SYMS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]   # 162 symbols, code = index (8 bits)


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _category(v):
    a = abs(int(v))
    return a.bit_length()


def _seg(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)


def encode(img, sampling=((1, 1), (1, 1), (1, 1)), qstep=4, restart=0):
    """img: [H, W, 3] uint8 RGB or [H, W] uint8 grey.  sampling: (h, v) per component.  qstep: flat quantiser step (1..255)."""
    img = np.asarray(img)
    grey = img.ndim == 2
    H, W = img.shape[:2]
    if grey:
        planes = [img.astype(np.float64)]
        sampling = sampling[:1]
    else:
        r, g, b = [img[..., i].astype(np.float64) for i in range(3)]
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128, 0.5 * r - 0.418688 * g - 0.081312 * b + 128]
    hmax = max(h for h, _ in sampling)
    vmax = max(v for _, v in sampling)
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    comps = []
    for p, (h, v) in zip(planes, sampling):
        sx, sy = hmax // h, vmax // v
        ph, pw = mcuy * 8 * vmax, mcux * 8 * hmax
        full = np.pad(p, ((0, ph - H), (0, pw - W)), mode="edge")
        sub = full.reshape(ph // sy, sy, pw // sx, sx).mean(axis=(1, 3))
        comps.append(sub - 128.0)
    out = bytearray(b"\xFF\xD8")
    out += _seg(0xDB, bytes([0]) + bytes([qstep] * 64))
    out += _seg(0xC0, bytes([8]) + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([len(comps)]) +
                b"".join(bytes([i + 1, (h << 4) | v, 0]) for i, (h, v) in enumerate(sampling)))
    out += _seg(0xC4, bytes([0x00]) + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12)))
    out += _seg(0xC4, bytes([0x10]) + bytes([0] * 7 + [len(AC_SYMS)] + [0] * 8) + bytes(AC_SYMS))
    if restart:
        out += _seg(0xDD, restart.to_bytes(2, "big"))
    out += _seg(0xDA, bytes([len(comps)]) + b"".join(bytes([i + 1, 0x00]) for i in range(len(comps))) + bytes([0, 63, 0]))
    ac_code = {s: i for i, s in enumerate(AC_SYMS)}
    bits = _Bits()
    pred = [0] * len(comps)
    count = 0
    rst = 0
    for my in range(mcuy):
        for mx in range(mcux):
            for ci, (c, (h, v)) in enumerate(zip(comps, sampling)):
                for by in range(v):
                    for bx in range(h):
                        y0, x0 = (my * v + by) * 8, (mx * h + bx) * 8
                        q = np.rint(dctn(c[y0:y0 + 8, x0:x0 + 8], norm="ortho") / qstep).astype(int).reshape(64)
                        zz = [int(np.clip(q[z], -1023, 1023)) for z in ZIGZAG]
                        diff = zz[0] - pred[ci]
                        pred[ci] = zz[0]
                        cat = _category(diff)
                        bits.put(cat, 4)
                        if cat:
                            bits.put(diff if diff > 0 else diff + (1 << cat) - 1, cat)
                        run = 0
                        last = max([k for k in range(1, 64) if zz[k]], default=0)
                        for k in range(1, last + 1):
                            if zz[k] == 0:
                                run += 1
                                continue
                            while run > 15:
                                bits.put(ac_code[0xF0], 8)
                                run -= 16
                            cat = _category(zz[k])
                            bits.put(ac_code[(run << 4) | cat], 8)
                            bits.put(zz[k] if zz[k] > 0 else zz[k] + (1 << cat) - 1, cat)
                            run = 0
                        if last < 63:
                            bits.put(ac_code[0x00], 8)
            count += 1
            if restart and count % restart == 0 and count < mcux * mcuy:
                bits.flush()
                out += bits.out
                out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                bits = _Bits()
                pred = [0] * len(comps)
    bits.flush()
    out += bits.out
    out += b"\xFF\xD9"
    return bytes(out)
