"""A minimal baseline JPEG ENCODER for tests (ITU-T T.81 sequential DCT, Huffman, one interleaved scan): it exists to
make inputs Pillow's encoder cannot (arbitrary per-component sampling factors such as 4:4:0 or mixed ones, component ids
'R','G','B', Adobe APP14 markers, 16-bit quantisation tables / SOF1, restart intervals).  Test infrastructure only."""
import struct

import numpy as np
from scipy.fft import dctn

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,
                   42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])

# any prefix-free canonical table will do: 12 DC categories at 4 bits; the 162 legal AC symbols at 8 and 9 bits
DC_SYMS = list(range(12))
DC_COUNTS = [0, 0, 0, 12] + [0] * 12
AC_SYMS = [0x00, 0xf0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
AC_COUNTS = [0] * 7 + [100, 62] + [0] * 7


def _codes(counts, syms):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            out[syms[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xff
            self.out.append(b)
            if b == 0xff:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)   # pad with ones


def _category(v):
    return int(abs(int(v))).bit_length()


def _amp(v, s):
    return v if v >= 0 else v + (1 << s) - 1


def encode(planes, sampling, width, height, quant=None, ids=(1, 2, 3), jfif=True, adobe=None, dqt16=False, sof=0xc0, restart=0):
    """planes[i]: the i-th component AT ITS OWN resolution (uint8, at least ceil(w*h_i/hmax) x ceil(h*v_i/vmax));
    sampling[i] = (h_i, v_i).  One quantisation table for all components (`quant`, 64 values in natural order)."""
    quant = np.full(64, 6, np.int64) if quant is None else np.asarray(quant, np.int64)
    n = len(planes)
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    dc, ac = _codes(DC_COUNTS, DC_SYMS), _codes(AC_COUNTS, AC_SYMS)
    out = bytearray(b"\xff\xd8")
    if jfif:
        out += b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"
    if adobe is not None:
        out += b"\xff\xee" + struct.pack(">H", 14) + b"Adobe\0" + struct.pack(">BHHB", 100, 0, 0, adobe)
    zq = quant[ZIGZAG]
    out += b"\xff\xdb" + (struct.pack(">HB", 2 + 1 + 128, 0x10) + b"".join(struct.pack(">H", int(q)) for q in zq) if dqt16
                          else struct.pack(">HB", 2 + 1 + 64, 0) + bytes(int(q) for q in zq))
    out += bytes([0xff, sof]) + struct.pack(">HBHHB", 8 + 3 * n, 8, height, width, n)
    for i in range(n):
        out += bytes([ids[i], (sampling[i][0] << 4) | sampling[i][1], 0])
    out += b"\xff\xc4" + struct.pack(">HB", 2 + 17 + len(DC_SYMS), 0x00) + bytes(DC_COUNTS) + bytes(DC_SYMS)
    out += b"\xff\xc4" + struct.pack(">HB", 2 + 17 + len(AC_SYMS), 0x10) + bytes(AC_COUNTS) + bytes(AC_SYMS)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * n, n)
    for i in range(n):
        out += bytes([ids[i], 0x00])
    out += bytes([0, 63, 0])
    # pad each plane to whole MCUs by edge replication
    padded = []
    for p, (h, v) in zip(planes, sampling):
        H, W = mcuy * v * 8, mcux * h * 8
        p = np.asarray(p, np.float64)
        padded.append(np.pad(p, ((0, H - p.shape[0]), (0, W - p.shape[1])), mode="edge"))
    bits, pred, count, rst = _Bits(), [0] * n, 0, 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart and count and count % restart == 0:
                bits.flush()
                out += bits.out + bytes([0xff, 0xd0 + (rst & 7)])
                bits, pred, rst = _Bits(), [0] * n, rst + 1
            count += 1
            for i, (h, v) in enumerate(sampling):
                for by in range(v):
                    for bx in range(h):
                        y0, x0 = (my * v + by) * 8, (mx * h + bx) * 8
                        coef = dctn(padded[i][y0:y0 + 8, x0:x0 + 8] - 128.0, norm="ortho")
                        q = np.rint(coef.reshape(64) / quant).astype(np.int64)[ZIGZAG]
                        d = int(q[0]) - pred[i]
                        pred[i] = int(q[0])
                        s = _category(d)
                        bits.put(*dc[s])
                        if s:
                            bits.put(_amp(d, s), s)
                        run = 0
                        last = int(np.max(np.nonzero(q)[0])) if np.any(q[1:]) else 0
                        for k in range(1, last + 1):
                            if q[k] == 0:
                                run += 1
                                continue
                            while run > 15:
                                bits.put(*ac[0xf0])
                                run -= 16
                            s = _category(q[k])
                            bits.put(*ac[(run << 4) | s])
                            bits.put(_amp(int(q[k]), s), s)
                            run = 0
                        if last < 63:
                            bits.put(*ac[0x00])
    bits.flush()
    return bytes(out + bits.out + b"\xff\xd9")


def subsample(plane, fx, fy):
    """box average by (fx, fy) with edge replication"""
    p = np.asarray(plane, np.float64)
    H, W = -(-p.shape[0] // fy) * fy, -(-p.shape[1] // fx) * fx
    p = np.pad(p, ((0, H - p.shape[0]), (0, W - p.shape[1])), mode="edge")
    return np.clip(np.rint(p.reshape(H // fy, fy, W // fx, fx).mean((1, 3))), 0, 255).astype(np.uint8)


def rgb_to_ycc(img):
    r, g, b = [img[..., k].astype(np.float64) for k in range(3)]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = 128 - 0.168736 * r - 0.331264 * g + 0.5 * b
    cr = 128 + 0.5 * r - 0.418688 * g - 0.081312 * b
    return [np.clip(np.rint(c), 0, 255).astype(np.uint8) for c in (y, cb, cr)]
