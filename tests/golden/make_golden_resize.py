"""tests/golden/ref_resize.npz: what the REFERENCE'S OWN stb_image_resize build (oracle/_ref/libptref.so,
ref_stbir_resize_512 = stbir_resize_uint8(..., 512, 512, 0, 3) as src/loader.cpp:224-232 calls it) makes of seeded images:
the CRC-32 of the 512 x 512 x 3 output and every 16th pixel of it, per case.  The inputs are regenerated from the seed by
the test (tests/test_gltf_loader.py::resize_case).  Run in the build container:  python tests/golden/make_golden_resize.py"""
import ctypes
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

SHAPES = [(1024, 1024), (256, 256), (300, 700), (64, 64), (512, 300), (300, 512), (513, 511), (1, 1), (2, 3), (2048, 16),
          (16, 2048), (511, 512), (1000, 37), (768, 1536)]


def resize_case(i):
    """seeded input number i: smooth-ish noise so that neighbouring texels correlate like a texture's"""
    shape = SHAPES[i]
    rng = np.random.default_rng(1000 + i)
    src = np.clip(rng.normal(128, 60, shape + (3,)), 0, 255).astype(np.uint8)
    if shape[0] * shape[1] > 100:
        src = (src.astype(np.float64) * 0.5 + np.roll(src, 3, 0) * 0.25 + np.roll(src, 5, 1) * 0.25).astype(np.uint8)
    return np.ascontiguousarray(src)


def main():
    R = ol.load_ref()
    assert R is not None and hasattr(R, "ref_stbir_resize_512"), "build oracle/_ref first: make -C oracle ref"
    out = {}
    for i, shape in enumerate(SHAPES):
        src = resize_case(i)
        ref = np.zeros((512, 512, 3), np.uint8)
        assert R.ref_stbir_resize_512(ctypes.c_void_p(src.ctypes.data), src.shape[1], src.shape[0], ctypes.c_void_p(ref.ctypes.data)) == 1
        out[f"crc_{i}"] = np.uint32(zlib.crc32(ref.tobytes()) & 0xffffffff)
        out[f"sub_{i}"] = ref[::16, ::16].copy()
        print(shape, hex(int(out[f"crc_{i}"])))
    np.savez_compressed(os.path.join(HERE, "ref_resize.npz"), **out)


if __name__ == "__main__":
    main()
