"""Bit-exact closest-hit contract: pt_trace_closest (HIP) vs intersect() of the oracle on identical rays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(n, seed, inside=True):
    rng = np.random.default_rng(seed)
    ro = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    if not inside:
        ro[: n // 2] = 0.0
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    # degenerate directions: exact zeros (inv_dir = inf, 0*inf = NaN in the slab test) and axis-aligned rays
    rd[:64, 0] = 0.0
    rd[64:128, 1] = 0.0
    rd[128:160] = [0, 0, -1]
    rd[160:192] = [0, -1, 0]
    ro[192:224] = [-2.0, 0.0, -6.0]  # origin exactly on a wall plane
    return ro, rd


def _check(pt, ol, scene, ro, rd):
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(scene)
        prim, th, kind, uv = c.trace_closest(ro, rd)
    finally:
        c.close()
    osc = ol.scene_from_desc(scene.desc())
    oprim, oth, okind, ouv = ol.trace_closest(osc, ro, rd)
    assert np.array_equal(kind, okind)
    assert np.array_equal(prim, oprim)
    assert np.array_equal(th, oth)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32))
    return kind


def test_cornell(pt, ol):
    ro, rd = _rays(200000, 1, inside=False)
    kind = _check(pt, ol, pt.Scene.cornell(), ro, rd)
    assert (kind == 1).mean() > 0.3


def test_cornell_spheres(pt, ol):
    ro, rd = _rays(200000, 2, inside=False)
    kind = _check(pt, ol, pt.Scene.cornell(spheres=True), ro, rd)
    assert (kind == 2).sum() > 100


def test_soup(pt, ol):
    ro, rd = _rays(300000, 3)
    _check(pt, ol, pt.Scene.soup(20000, seed=11, size=0.3), ro, rd)


def test_bunny(pt, ol):
    ro, rd = _rays(300000, 4, inside=False)
    _check(pt, ol, pt.Scene.bunny(spheres=True), ro, rd)


def test_ragged_sizes(pt, ol):
    sc = pt.Scene.soup(3000, seed=5)
    for n in (1, 63, 64, 65, 1000):
        ro, rd = _rays(max(n, 256), 10 + n)
        _check(pt, ol, sc, ro[:n], rd[:n])
