"""Seeded synthetic inputs shared by tests/golden/make_golden.py (which feeds them to the reference's own
translation units) and by the tests (which feed the same inputs to the build's host code / oracle / HIP path)."""
import hashlib

import numpy as np


def digest64(data: bytes) -> int:
    """First 8 bytes of SHA-256 as an integer: a stable digest for byte-equality of large outputs."""
    return int.from_bytes(hashlib.sha256(data).digest()[:8], "little")


def lcg_mesh(n_tris, seed, extent=4.0, size=0.35, snap=0.0):
    """n packed 64-byte triangles (as u32[n,16]): random small triangles in a cube; uv/normal words arbitrary.
    snap > 0 quantises vertex coordinates to that grid (equal centroids, flat boxes)."""
    rng = np.random.default_rng(seed)  # PCG64: deterministic across numpy versions for these calls
    c = rng.uniform(-extent, extent, (n_tris, 1, 3))
    v = (c + rng.uniform(-size, size, (n_tris, 3, 3))).astype(np.float32)
    if snap > 0:
        v = (np.round(v / snap) * snap).astype(np.float32)
    t = np.zeros((n_tris, 16), np.uint32)
    f = t.view(np.float32)
    for k in range(3):
        f[:, 4 * k: 4 * k + 3] = v[:, k, :]
    t[:, 3] = rng.integers(0, 2**32, n_tris, dtype=np.uint64).astype(np.uint32) & 0x3bff3bff
    t[:, 12:15] = rng.integers(0, 2**32, (n_tris, 3), dtype=np.uint64).astype(np.uint32) & 0x3bff3bff
    t[:, 15] = 0
    return t


def fat_triangles(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-3, 3, (n, 9)).astype(np.float32)
    uv = rng.uniform(-2, 3, (n, 6)).astype(np.float32)
    nr = rng.normal(size=(n, 3, 3))
    nr /= np.linalg.norm(nr, axis=2, keepdims=True)
    nr = nr.astype(np.float32).reshape(n, 9)
    mat = rng.integers(0, 200, n).astype(np.uint32)
    # special values: axis normals, zeros, fp16 boundary uvs, half-way cases for the fp32->fp16 rounding
    nr[0] = [0, 0, 1, 0, 0, -1, 1, 0, 0]
    nr[1] = [0, 1, 0, 0, -1, 0, -1, 0, 0]
    uv[2] = [0, 1, 0.5, 0.25, 65504.0, -65504.0]
    uv[3] = [1e-8, 6.1e-5, 5.96e-8, 2.98e-8, 1.0009765625, 1.00048828125]  # subnormal halves, exact ties
    uv[4] = [1.0 + 2.0**-11, 1.0 + 3 * 2.0**-11, 2049.0, 2051.0, 0.1, 0.2]
    uv[5] = [70000.0, -70000.0, np.inf, -np.inf, 65520.0, 65519.0]
    return v, uv, nr, mat


def materials_in():
    f = lambda *a: np.array(a, np.float32)
    return [(f(0.73, 0.73, 0.73), f(0, 0, 0), f(0, 0, 0), 0.0, -1, 0),
            (f(0.65, 0.05, 0.05), f(0, 0, 0), f(0, 0, 0), 0.35, 3, 1),
            (f(0, 0, 0), f(0.944, 0.776, 0.373), f(0, 0, 0), 0.0, -1, 2),      # metal roughness clamped to 0.01
            (f(0, 0, 0), f(0.944, 0.776, 0.373), f(0, 0, 0), 0.005, -1, 2),
            (f(0, 0, 0), f(0.944, 0.776, 0.373), f(0, 0, 0), 0.2, 7, 2),
            (f(1, 1, 1), f(0, 0, 0), f(17, 12, 4), 0.0, -1, -1),
            (f(0.1, 0.2, 0.3), f(0.4, 0.5, 0.6), f(0.7, 0.8, 0.9), 2.0, 0, 0)]


def spheres_in():
    f = lambda *a: np.array(a, np.float32)
    return [(f(-0.75, -1.45, -4.4), 0.3, 0), (f(-0.25, -1.45, -4.4), 0.3, 1), (f(0, 0, 0), 1e-3, 255), (f(1e6, -1e6, 3), 1e4, 4000000000)]
