"""MIS frames: HIP wavefront path vs the oracle restatement, same scenes, seeds and spp."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render_both(pt, ol, scene, w, h, nframes, bounces=5, flags=0, env=None):
    cam = pt.default_camera()
    seeds = pt.frame_seeds(0, nframes)
    c = pt.PathTracer(w, h, flags=flags)
    try:
        c.upload_scene(scene, env_rgb8=env)
        c.set_camera(cam)
        c.render(0, seeds, bounces)
        img = c.read_framebuffer()
        cnt = c.counters()
    finally:
        c.close()
    osc = ol.scene_from_desc(scene.desc(), env_rgb8=env)
    ref, ocnt = ol.render(osc, cam, w, h, seeds, bounces=bounces)
    return img, ref, cnt, ocnt


def _report(img, ref):
    d = np.abs(img[..., :3] - ref[..., :3])
    same = (img.view(np.uint32) == ref.view(np.uint32)).all(axis=2).mean()
    rmse = float(np.sqrt(np.mean((img[..., :3] - ref[..., :3]) ** 2)))
    return same, rmse, float(d.max())


@pytest.mark.parametrize("name", ["cornell", "cornell_spheres", "bunny_on", "bunny_ggx"])
def test_frame_parity(pt, ol, name):
    scene = {"cornell": lambda: pt.Scene.cornell(), "cornell_spheres": lambda: pt.Scene.cornell(spheres=True),
             "bunny_on": lambda: pt.Scene.bunny(freq=12, target_tris=0),
             "bunny_ggx": lambda: pt.Scene.bunny(freq=12, target_tris=0, material=pt.MAT_SPECULAR_METAL, roughness=0.2, spheres=True)}[name]()
    img, ref, cnt, ocnt = _render_both(pt, ol, scene, 64, 64, 4, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
    same, rmse, dmax = _report(img, ref)
    print(name, "bit-identical pixels:", same, "rmse", rmse, "max", dmax, cnt, ocnt)
    # PCG streams identical per pixel/sample + identical arithmetic contract => identical pixels
    assert same == 1.0
    for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce"):
        assert cnt[k] == ocnt[k], k


def test_retiring_dead_paths_keeps_the_image(pt, ol):
    scene = pt.Scene.cornell(spheres=True)
    a, ref, cnt_a, ocnt = _render_both(pt, ol, scene, 64, 64, 3, flags=0)
    same, rmse, dmax = _report(a, ref)
    assert same == 1.0
    assert cnt_a["rays_bounce"] <= ocnt["rays_bounce"]


def test_batch_size_does_not_change_the_image(pt):
    """Frames in flight per launch set (pt_config.max_batch) is a scheduling choice: samples are blended in frame order."""
    scene = pt.Scene.bunny(freq=10, target_tris=0, spheres=True)
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 7)
    imgs = []
    for mb, pipes in ((1, 1), (3, 2), (8, 1), (16, 2), (2, 4)):
        c = pt.PathTracer(72, 40, max_batch=mb, pipelines=pipes)
        try:
            c.upload_scene(scene)
            c.set_camera(cam)
            c.render(0, seeds, 5)
            imgs.append(c.read_framebuffer())
        finally:
            c.close()
    c = pt.PathTracer(72, 40)
    try:
        c.upload_scene(scene)
        c.set_camera(cam)
        for f in range(7):
            c.render_frame(int(seeds[f]), f, 5)   # the reference's own call pattern: one dispatch per frame
        imgs.append(c.read_framebuffer())
        c.render(7, pt.frame_seeds(7, 2), 5)      # and resuming an accumulation
        more = c.read_framebuffer()
    finally:
        c.close()
    for im in imgs[1:]:
        assert np.array_equal(imgs[0].view(np.uint32), im.view(np.uint32))
    assert not np.array_equal(more, imgs[0])


def _render(pt, scene, w, h, n, flags):
    c = pt.PathTracer(w, h, flags=flags)
    try:
        c.upload_scene(scene)
        c.set_camera(pt.default_camera())
        c.render(0, pt.frame_seeds(0, n), 5)
        return c.read_framebuffer(), c.counters()
    finally:
        c.close()


@pytest.mark.parametrize("name", ["cornell_spheres", "bunny"])
def test_shadow_early_out_is_exact(pt, name):
    """Default shadow traversal stops at the first hit that already decides the reference's compare: same pixels, same
    ray counts, less traversal work than going on to the closest hit."""
    scene = pt.Scene.cornell(spheres=True) if name == "cornell_spheres" else pt.Scene.bunny(freq=16, target_tris=0, spheres=True)
    full, cf = _render(pt, scene, 96, 64, 4, pt.PT_FLAG_COUNT_TRAVERSAL | pt.PT_FLAG_SHADOW_FULL_TRAVERSAL)
    fast, ce = _render(pt, scene, 96, 64, 4, pt.PT_FLAG_COUNT_TRAVERSAL)
    assert np.array_equal(full.view(np.uint32), fast.view(np.uint32))
    for k in ("rays_primary", "rays_shadow", "rays_bounce"):
        assert cf[k] == ce[k]
    assert ce["node_steps"] < cf["node_steps"] and ce["tri_tests"] <= cf["tri_tests"]
    print(name, "node steps", cf["node_steps"], "->", ce["node_steps"], "tri tests", cf["tri_tests"], "->", ce["tri_tests"])


def test_anyhit_mode_is_close(pt):
    """PT_FLAG_SHADOW_ANYHIT drops the one-sided quirk of the shadow compare: not bit-exact, but the same picture.
    Stated tolerance: image RMSE <= 3 % of the mean radiance at 32 spp, mean radiance within 2 %."""
    scene = pt.Scene.bunny(freq=16, target_tris=0)
    ref, cr = _render(pt, scene, 96, 64, 32, pt.PT_FLAG_COUNT_TRAVERSAL)
    any_, ca = _render(pt, scene, 96, 64, 32, pt.PT_FLAG_COUNT_TRAVERSAL | pt.PT_FLAG_SHADOW_ANYHIT)
    mean = float(ref[..., :3].mean())
    rmse = float(np.sqrt(np.mean((ref[..., :3] - any_[..., :3]) ** 2)))
    print("anyhit: mean", mean, float(any_[..., :3].mean()), "rmse", rmse, "steps", cr["node_steps"], "->", ca["node_steps"])
    assert abs(float(any_[..., :3].mean()) - mean) <= 0.02 * mean
    assert rmse <= 0.03 * mean
    assert ca["node_steps"] < cr["node_steps"]
    assert float(any_[..., :3].mean()) <= mean * 1.0001  # any-hit can only remove light the quirk let through


def test_checkpoint_resume(pt, tmp_path):
    """Accumulation state = image + next frame number: saved as PFM, restored into a fresh context, continued."""
    scene = pt.Scene.cornell(spheres=True)
    seeds = pt.frame_seeds(0, 10)
    whole, _ = _render(pt, scene, 64, 48, 10, 0)
    c = pt.PathTracer(64, 48)
    try:
        c.upload_scene(scene)
        c.set_camera(pt.default_camera())
        c.render(0, seeds[:6], 5)
        pt.write_pfm(str(tmp_path / "ckpt.pfm"), c.read_framebuffer())
    finally:
        c.close()
    c = pt.PathTracer(64, 48)
    try:
        c.upload_scene(scene)
        c.set_camera(pt.default_camera())
        c.write_framebuffer(pt.read_pfm(str(tmp_path / "ckpt.pfm"), 64, 48))
        c.render(6, seeds[6:], 5)
        resumed = c.read_framebuffer()
    finally:
        c.close()
    assert np.array_equal(whole.view(np.uint32), resumed.view(np.uint32))


def _textured_scene(pt):
    """A lit box whose floor, back wall and a GGX block sample three layers of the 512x512 RGB8 texture array
    (calc_BRDF's texture() path, framebuffer.comp:599-603; GL_REPEAT + GL_LINEAR: uvs run past [0,1])."""
    s = pt.Scene()
    white = s.add_material([0.73, 0.73, 0.73], [0, 0, 0], [0, 0, 0])
    floor = s.add_material([0.9, 0.9, 0.9], [0, 0, 0], [0, 0, 0], roughness=0.2, tex_index=0, mat_type=pt.MAT_OREN_NAYAR)
    back = s.add_material([1.0, 1.0, 1.0], [0, 0, 0], [0, 0, 0], tex_index=1, mat_type=pt.MAT_LAMBERTIAN)
    metal = s.add_material([0, 0, 0], [0.944, 0.776, 0.373], [0, 0, 0], roughness=0.15, tex_index=2, mat_type=pt.MAT_SPECULAR_METAL)
    light = s.add_material([0, 0, 0], [0, 0, 0], [17, 12, 4], mat_type=pt.MAT_LIGHT)

    def quad(a, b, c, d, mat, uvscale=1.0):
        a, b, c, d = (np.array(p, np.float32) for p in (a, b, c, d))
        n = np.cross(b - a, c - a)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uvscale - 0.25
        v = np.stack([np.concatenate([a, b, c]), np.concatenate([a, c, d])])
        nn = np.tile(n, (2, 3))
        uvs = np.stack([np.concatenate([uv[0], uv[1], uv[2]]), np.concatenate([uv[0], uv[2], uv[3]])])
        s.add_triangles(v, nn, [mat, mat], uvs)

    x0, x1, y0, y1, z0, z1 = -2.0, 2.0, -2.0, 2.0, -8.0, -4.0
    quad([x0, y0, z1], [x1, y0, z1], [x1, y0, z0], [x0, y0, z0], floor, 3.0)
    quad([x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], back, 2.0)
    quad([x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1], white)
    quad([x0, y0, z1], [x0, y0, z0], [x0, y1, z0], [x0, y1, z1], white)
    quad([x1, y0, z0], [x1, y0, z1], [x1, y1, z1], [x1, y1, z0], white)
    quad([-0.5, 1.99, -6.5], [0.5, 1.99, -6.5], [0.5, 1.99, -5.5], [-0.5, 1.99, -5.5], light)
    quad([-1.2, -1.0, -5.5], [0.2, -1.0, -5.2], [0.2, -1.0, -6.6], [-1.2, -1.0, -6.9], metal, 1.5)   # tilted-ish slab top
    quad([-1.2, -2.0, -5.5], [0.2, -2.0, -5.2], [0.2, -1.0, -5.2], [-1.2, -1.0, -5.5], metal, 1.0)   # its front
    s.finalize()
    rng = np.random.default_rng(12)
    tex = rng.integers(0, 256, (3, 512, 512, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:512, 0:512]
    tex[0] = np.stack([((xx // 32 + yy // 32) % 2) * 200 + 30, xx // 2, yy // 2], 2).astype(np.uint8)  # checker + ramps
    return s, tex


def test_textured_materials(pt, ol):
    scene, tex = _textured_scene(pt)
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 4)
    c = pt.PathTracer(96, 96, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
    try:
        c.upload_scene(scene, tex_rgb8=tex)
        c.set_camera(cam)
        c.render(0, seeds, 5)
        img = c.read_framebuffer()
    finally:
        c.close()
    ref, _ = ol.render(ol.scene_from_desc(scene.desc(), tex_rgb8=tex), cam, 96, 96, seeds)
    untextured, _ = ol.render(ol.scene_from_desc(scene.desc()), cam, 96, 96, seeds)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert not np.array_equal(ref, untextured)  # the layers are really being sampled


@pytest.mark.parametrize("sampler", ["clamp_clamp_linear", "mirror_repeat_nearest", "repeat_mirror_linear"])
def test_texture_sampler_state(pt, ol, sampler):
    """Wrap modes / magnification filter of the texture array (the glTF sampler the loader applies, loader.cpp:242-251):
    uvs run from -0.25 to 2.75, so every mode gives a different image, each bit-identical to the oracle's."""
    modes = {"clamp_clamp_linear": (pt.GL_CLAMP_TO_EDGE, pt.GL_CLAMP_TO_EDGE, pt.GL_LINEAR),
             "mirror_repeat_nearest": (pt.GL_MIRRORED_REPEAT, pt.GL_REPEAT, pt.GL_NEAREST),
             "repeat_mirror_linear": (pt.GL_REPEAT, pt.GL_MIRRORED_REPEAT, pt.GL_LINEAR)}[sampler]
    scene, tex = _textured_scene(pt)
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 3)
    c = pt.PathTracer(96, 96, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
    try:
        c.upload_scene(scene, tex_rgb8=tex, tex_sampler=modes)
        c.set_camera(cam)
        c.render(0, seeds, 5)
        img = c.read_framebuffer()
        with pytest.raises(pt.PtError):
            c.upload_scene(scene, tex_rgb8=tex, tex_sampler=(1234, pt.GL_REPEAT, pt.GL_LINEAR))
    finally:
        c.close()
    ref, _ = ol.render(ol.scene_from_desc(scene.desc(), tex_rgb8=tex, tex_sampler=modes), cam, 96, 96, seeds)
    default, _ = ol.render(ol.scene_from_desc(scene.desc(), tex_rgb8=tex), cam, 96, 96, seeds)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert not np.array_equal(ref, default)


@pytest.mark.parametrize("est", ["nee", "brdf"])
@pytest.mark.parametrize("name", ["cornell_spheres", "bunny_ggx_env"])
def test_alternative_estimators(pt, ol, est, name):
    """estimator_path_tracing_nee / _BRDF (framebuffer.comp:725-867 / 647-693; the reference selects them by editing
    fc:1109-1111): HIP wavefront stages vs the oracle's restatement, bit for bit, incl. ray counts."""
    from test_oracle import golden_env
    if name == "cornell_spheres":
        scene, env = pt.Scene.cornell(spheres=True), None
    else:
        scene, env = pt.Scene.bunny(freq=12, target_tris=0, material=pt.MAT_SPECULAR_METAL, roughness=0.2, spheres=True), golden_env()
    which, oflag = (pt.PT_ESTIMATOR_NEE, ol.FLAG_EST_NEE) if est == "nee" else (pt.PT_ESTIMATOR_BRDF, ol.FLAG_EST_BRDF)
    cam = pt.default_camera()
    for bounces, n in ((5, 4), (1, 2), (0, 1)):
        seeds = pt.frame_seeds(0, n)
        c = pt.PathTracer(64, 64, flags=pt.PT_FLAG_TRACE_DEAD_PATHS, max_batch=3)
        try:
            c.upload_scene(scene, env_rgb8=env)
            c.set_camera(cam)
            c.set_estimator(which)
            c.render(0, seeds, bounces)
            img, cnt = c.read_framebuffer(), c.counters()
        finally:
            c.close()
        ref, ocnt = ol.render(ol.scene_from_desc(scene.desc(), env_rgb8=env), cam, 64, 64, seeds, bounces=bounces, flags=oflag)
        bad = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        assert bad == 0, f"{est} bounces {bounces}: {bad} pixels differ"
        for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce"):
            assert cnt[k] == ocnt[k], (est, bounces, k, cnt[k], ocnt[k])


def test_estimators_agree_in_expectation(pt):
    """Three estimators of the same integral... almost: the BRDF estimator runs one path segment fewer (its loop makes
    `bounces` intersect() calls, MIS makes bounces + 1), NEE and MIS see the same depth.  NEE vs MIS mean radiance within
    3 % at 64 spp (they differ by MIS's reference quirks 9 and 16, not by noise alone)."""
    scene = pt.Scene.cornell()
    means = {}
    for which in (pt.PT_ESTIMATOR_MIS, pt.PT_ESTIMATOR_NEE):
        c = pt.PathTracer(64, 64)
        try:
            c.upload_scene(scene)
            c.set_camera(pt.default_camera())
            c.set_estimator(which)
            c.render(0, pt.frame_seeds(0, 64), 5)
            means[which] = float(c.read_framebuffer()[..., :3].mean())
        finally:
            c.close()
    assert abs(means[pt.PT_ESTIMATOR_MIS] - means[pt.PT_ESTIMATOR_NEE]) < 0.03 * means[pt.PT_ESTIMATOR_MIS], means
