"""Test infrastructure: a zlib stream of ONE final fixed-Huffman block (RFC 1951 3.2.6) from an explicit token list -- what the reference's writer
emits (stb_image_write.h:733-736), with the tokens chosen by the test instead of a match finder: runs that copy themselves, chains of short near
matches, the longest distance, invalid codes."""
import zlib

_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
_DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def raw(self, value, nbits):          # LSB first
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, nbits):          # Huffman codes go MSB first
        self.raw(int(format(code, "0%db" % nbits)[::-1], 2), nbits)

    def symbol(self, s):                  # lit / len alphabet of the fixed code
        if s < 144:
            self.code(0x30 + s, 8)
        elif s < 256:
            self.code(0x190 + s - 144, 9)
        elif s < 280:
            self.code(s - 256, 7)
        else:
            self.code(0xC0 + s - 280, 8)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def fixed_block(tokens, end=True, header=(1, 1)):
    """tokens: ints (literals), (length, distance) pairs, or ("sym", s) / ("dist", code) for raw symbols -> the deflate bytes."""
    b = _Bits()
    b.raw(header[0], 1)
    b.raw(header[1], 2)
    for t in tokens:
        if isinstance(t, int):
            b.symbol(t)
        elif t[0] == "sym":
            b.symbol(t[1])
        elif t[0] == "dist":
            b.code(t[1], 5)
        else:
            length, dist = t
            li = max(i for i in range(29) if _LEN_BASE[i] <= length and (i < 28 or length == 258))
            if length == 258:
                li = 28
            b.symbol(257 + li)
            b.raw(length - _LEN_BASE[li], _LEN_EXTRA[li])
            di = max(i for i in range(30) if _DIST_BASE[i] <= dist)
            b.code(di, 5)
            b.raw(dist - _DIST_BASE[di], _DIST_EXTRA[di])
    if end:
        b.symbol(256)
    return b.done()


def apply_tokens(tokens):
    out = bytearray()
    for t in tokens:
        if isinstance(t, int):
            out.append(t)
        else:
            length, dist = t
            for _ in range(length):
                out.append(out[-dist])
    return bytes(out)


def zlib_stream(tokens, **kw):
    try:
        raw = apply_tokens(tokens)      # the trailer nobody checks (the reference's inflater does not: stb_image.h:3846)
    except (IndexError, TypeError, ValueError):
        raw = b""                       # corrupt on purpose
    return b"\x78\x01" + fixed_block(tokens, **kw) + zlib.adler32(raw).to_bytes(4, "big")
