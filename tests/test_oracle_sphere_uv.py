"""The sphere branch of intersect() (framebuffer.comp:447-451) writes t, normal, mat_index, object_index and object_type
but NOT uvs: a sphere hit keeps the uvs of the last triangle the BVH pass accepted -- the closest triangle along the
ray, because every accepted hit lowers tmax -- and calc_BRDF (fc:599-603) textures the sphere's material with them.
Oracle side of that rule, pinned by an independent float64 brute force over the triangles as they were added."""
import numpy as np

import scene_textured_spheres as sts


def _closest_triangle_uv(fat, ro, rd):
    """float64 Moller-Trumbore over all triangles; returns (t, interpolated uv) of the closest hit or (inf, 0)."""
    v = fat["v"].astype(np.float64).reshape(-1, 3, 3)
    uv = fat["uv"].astype(np.float64).reshape(-1, 3, 2)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    p = np.cross(rd, e2)
    det = (e1 * p).sum(1)
    ok = np.abs(det) > 1e-9
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv = ro - v[:, 0]
    u = (tv * p).sum(1) * inv
    q = np.cross(tv, e1)
    w = (rd * q).sum(1) * inv
    t = (e2 * q).sum(1) * inv
    ok &= (u >= 0) & (w >= 0) & (u + w <= 1) & (t > 1e-3)
    if not ok.any():
        return np.inf, np.zeros(2)
    t = np.where(ok, t, np.inf)
    i = int(np.argmin(t))
    return t[i], (1 - u[i] - w[i]) * uv[i, 0] + u[i] * uv[i, 1] + w[i] * uv[i, 2]


def _rays_at_spheres(n_per, seed):
    rng = np.random.default_rng(seed)
    ro, rd = [], []
    for (c, r) in sts.SPHERES:
        c = np.array(c)
        for _ in range(n_per):
            o = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(-4.6, -4.1)])
            tgt = c + rng.uniform(-0.5, 0.5, 3) * r
            d = tgt - o
            ro.append(o)
            rd.append(d / np.linalg.norm(d))
    return np.array(ro, np.float32), np.array(rd, np.float32)


def test_sphere_hit_keeps_the_closest_triangles_uvs(pt, ol):
    scene, _ = sts.build(pt)
    osc = ol.scene_from_desc(scene.desc())
    fat = scene.fat_triangles()
    ro, rd = _rays_at_spheres(64, 5)
    prim, th, kind, uv = ol.trace_closest(osc, ro, rd)
    assert (kind == 2).mean() > 0.9
    checked = 0
    for i in np.nonzero(kind == 2)[0]:
        t_tri, uv_tri = _closest_triangle_uv(fat, ro[i].astype(np.float64), rd[i].astype(np.float64))
        assert np.isfinite(t_tri)  # the box is closed behind every sphere as seen from the open front
        # fp16 uvs (2^-11 relative), fp16-quantised traversal ray (hit point moves by ~5e-4 * t)
        assert np.allclose(uv[i], uv_tri, atol=0.02), (i, uv[i], uv_tri)
        checked += 1
    assert checked > 200
    assert np.abs(uv[kind == 2]).max() > 0.5  # really the walls' uvs, not zeros


def test_sphere_hit_without_a_triangle_behind_has_zero_uvs(pt, ol):
    scene, _ = sts.build(pt)
    osc = ol.scene_from_desc(scene.desc())
    # from behind the first sphere towards the open front of the box: the sphere, then nothing
    ro = np.array([[0.0, 0.2, -7.7]], np.float32)
    rd = np.array([[0.0, 0.0, 1.0]], np.float32)
    prim, th, kind, uv = ol.trace_closest(osc, ro, rd)
    assert kind[0] == 2 and prim[0] == 0
    assert np.array_equal(uv[0].view(np.uint32), np.zeros(2, np.uint32))


def test_untextured_walls_uvs_show_through_textured_spheres(pt, ol):
    """Moving the uvs of walls whose own material has NO texture changes the image -- only through the spheres in front of
    them.  With 1 bounce a pixel whose camera ray hits a triangle never uses any other primitive's uvs, so every pixel that
    changes is a sphere pixel."""
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 1)
    imgs = []
    for shift in (0.0, 0.37):
        scene, tex = sts.build(pt, wall_uv_shift=shift)
        img, _ = ol.render(ol.scene_from_desc(scene.desc(), tex_rgb8=tex), cam, 128, 128, seeds, bounces=1)
        imgs.append(img)
    changed = (imgs[0].view(np.uint32) != imgs[1].view(np.uint32)).any(axis=2)
    assert changed.sum() > 20
    # and with the sphere materials' textures taken away nothing changes at all
    plain = []
    for shift in (0.0, 0.37):
        scene, _ = sts.build(pt, wall_uv_shift=shift)
        img, _ = ol.render(ol.scene_from_desc(scene.desc()), cam, 128, 128, seeds, bounces=1)
        plain.append(img)
    assert np.array_equal(plain[0].view(np.uint32), plain[1].view(np.uint32))
