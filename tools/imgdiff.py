#!/usr/bin/env python3
"""Compare two PFM images (pth_write_pfm): RMSE, max abs / relative error, share of bit-identical pixels; optional
error map as PPM.   usage: tools/imgdiff.py a.pfm b.pfm [--map err.ppm] [--rel-floor 1e-3]"""
import argparse
import sys

import numpy as np


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF", "not a colour PFM"
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(w * h * 12), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--map")
    ap.add_argument("--rel-floor", type=float, default=1e-3)
    args = ap.parse_args()
    a, b = read_pfm(args.a), read_pfm(args.b)
    if a.shape != b.shape:
        sys.exit(f"shapes differ: {a.shape} vs {b.shape}")
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(a), args.rel_floor)
    same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=2).mean()
    print(f"pixels {a.shape[1]}x{a.shape[0]}  bit-identical {same * 100:.4f} %  rmse {np.sqrt(np.mean(d ** 2)):.6g}  "
          f"max abs {d.max():.6g}  max rel {rel.max():.6g}  mean a {a.mean():.6g}  mean b {b.mean():.6g}")
    if args.map:
        e = np.clip(rel.max(axis=2) / max(rel.max(), 1e-30), 0, 1)
        img = (np.stack([e, e, e], 2)[::-1] * 255).astype(np.uint8)
        with open(args.map, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
            f.write(img.tobytes())


if __name__ == "__main__":
    main()
