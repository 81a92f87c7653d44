#!/usr/bin/env python3
"""Regenerates tests/golden/gltf/*.glb and tests/golden/ref_gltf.npz.  BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_golden_gltf.py

The .glb files are written by tests/gltf_writer.py (own code, synthetic content, texture-less).  ref_gltf.npz holds what
the REFERENCE'S OWN LoadGLTF + Model (src/loader.cpp, src/scene/model.cpp, compiled from /root/reference into
oracle/_ref/libptref.so with the reference's vendored cgltf / stb / glad) produce for each file, twice: as loaded, and
after the scene set-up of src/main.cpp:26-29 (Translate(0,-2,-6), Rotate(0,-90 deg,0), Scale(2), ApplyModelMatrixToTris).
Data only: inputs and the reference's output bytes."""
import ctypes
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gltf_writer as gw  # noqa: E402
import oracle_lib as ol  # noqa: E402

P = ctypes.c_void_p
MAIN_OPS = np.array([[0, 0.0, -2.0, -6.0], [1, 0.0, np.float32(math.radians(-90.0)), 0.0], [2, 2.0, 2.0, 2.0], [3, 0, 0, 0]], np.float32)


def ptr(a):
    return a.ctypes.data_as(P)


def ref_load(R, path, ops):
    max_t, max_m = 20000, 64
    fat = np.zeros((max_t, 25), np.float32)
    mi = np.zeros(max_t, np.uint32)
    t64 = np.zeros((max_t, 16), np.uint32)
    m48 = np.zeros((max_m, 12), np.uint32)
    nm = ctypes.c_uint32()
    ops = np.ascontiguousarray(ops, np.float32).reshape(-1, 4)
    R.ref_load_gltf.restype = ctypes.c_int
    n = R.ref_load_gltf(path.encode(), ptr(ops), ops.shape[0], max_t, ptr(fat), ptr(mi), ptr(t64), max_m, ptr(m48), ctypes.byref(nm))
    assert n >= 0, (path, n)
    return fat[:n, :24].copy(), mi[:n].copy(), t64[:n].copy(), m48[: nm.value].copy()


def main():
    R = ol.load_ref()
    assert R is not None, "needs /root/reference (build container)"
    os.makedirs(os.path.join(HERE, "gltf"), exist_ok=True)
    out = {}
    for name, blob in gw.build_cases().items():
        path = os.path.join(HERE, "gltf", name + ".glb")
        with open(path, "wb") as f:
            f.write(blob)
        for tag, ops in (("loaded", np.zeros((0, 4), np.float32)), ("main", MAIN_OPS)):
            fat, mi, t64, m48 = ref_load(R, path, ops)
            out[f"{name}_{tag}_fat"], out[f"{name}_{tag}_mat"], out[f"{name}_{tag}_tris"], out[f"{name}_{tag}_mats"] = fat, mi, t64, m48
        print(name, "triangles", out[f"{name}_loaded_fat"].shape[0], "materials", out[f"{name}_loaded_mats"].shape[0])
    out["main_ops"] = MAIN_OPS
    np.savez_compressed(os.path.join(HERE, "ref_gltf.npz"), **out)
    print("ref_gltf.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
