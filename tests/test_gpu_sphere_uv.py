"""Textured SPHERE materials: a sphere hit keeps the uvs of the last accepted triangle (framebuffer.comp:447-451 does not
write uvs; calc_BRDF samples the texture with them, fc:599-603).  HIP vs oracle, bit for bit: the hit records, and whole
frames of all three estimators."""
import numpy as np
import pytest

import scene_textured_spheres as sts
from test_oracle_sphere_uv import _rays_at_spheres

pytestmark = pytest.mark.gpu


def test_trace_records_carry_the_uvs(pt, ol):
    scene, _ = sts.build(pt)
    rng = np.random.default_rng(3)
    n = 100000
    ro = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ro2, rd2 = _rays_at_spheres(2000, 9)
    ro, rd = np.concatenate([ro, ro2]), np.concatenate([rd, rd2])
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(scene)
        prim, th, kind, uv = c.trace_closest(ro, rd)
    finally:
        c.close()
    oprim, oth, okind, ouv = ol.trace_closest(ol.scene_from_desc(scene.desc()), ro, rd)
    assert np.array_equal(kind, okind) and np.array_equal(prim, oprim) and np.array_equal(th, oth)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32))
    sph = kind == 2
    assert sph.sum() > 5000
    assert (np.abs(uv[sph]).max(axis=1) > 0).mean() > 0.5   # most sphere hits have a triangle behind them: carried uvs
    assert (np.abs(uv[sph]).max(axis=1) == 0).sum() > 0      # and some (towards the open front) have none: zeros


@pytest.mark.parametrize("est", ["mis", "nee", "brdf"])
def test_textured_spheres_frames(pt, ol, est):
    scene, tex = sts.build(pt)
    which, oflag = {"mis": (pt.PT_ESTIMATOR_MIS, 0), "nee": (pt.PT_ESTIMATOR_NEE, ol.FLAG_EST_NEE),
                    "brdf": (pt.PT_ESTIMATOR_BRDF, ol.FLAG_EST_BRDF)}[est]
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 4)
    c = pt.PathTracer(96, 96, flags=pt.PT_FLAG_TRACE_DEAD_PATHS, max_batch=3)
    try:
        c.upload_scene(scene, tex_rgb8=tex)
        c.set_camera(cam)
        c.set_estimator(which)
        c.render(0, seeds, 5)
        img, cnt = c.read_framebuffer(), c.counters()
    finally:
        c.close()
    ref, ocnt = ol.render(ol.scene_from_desc(scene.desc(), tex_rgb8=tex), cam, 96, 96, seeds, flags=oflag)
    bad = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    assert bad == 0, f"{est}: {bad} pixels differ"
    for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce"):
        assert cnt[k] == ocnt[k], (est, k)
    # the carried uvs are really in play: the oracle's image changes when only the untextured walls' uvs move
    scene2, _ = sts.build(pt, wall_uv_shift=0.37)
    ref2, _ = ol.render(ol.scene_from_desc(scene2.desc(), tex_rgb8=tex), cam, 96, 96, seeds, flags=oflag)
    assert not np.array_equal(ref.view(np.uint32), ref2.view(np.uint32))


def test_default_flags_and_anyhit_survive_textured_spheres(pt, ol):
    """Default flags (dead paths retired, exact shadow early-out): same image as the oracle; the shadow segment's records
    of sphere occluders take the early paths of the refill."""
    scene, tex = sts.build(pt)
    cam, seeds = pt.default_camera(), pt.frame_seeds(3, 3)
    c = pt.PathTracer(64, 64)
    try:
        c.upload_scene(scene, tex_rgb8=tex)
        c.set_camera(cam)
        c.render(3, seeds, 5)
        img = c.read_framebuffer()
    finally:
        c.close()
    ref = np.zeros((64, 64, 4), np.float32)
    ref, _ = ol.render(ol.scene_from_desc(scene.desc(), tex_rgb8=tex), cam, 64, 64, seeds, first_frame=3, rgba=ref)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
