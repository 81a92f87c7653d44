"""Edge cases of the path: a scene without any light (the only case in which the shader's GGX-without-NEE emission
add, framebuffer.comp:1071-1077, and its num_lights == 0 guards are live), empty inputs, the smallest image, zero
bounces, a single-triangle scene."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env(seed=5, w=64, h=32):
    rng = np.random.default_rng(seed)
    env = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    env[: h // 3] = [250, 240, 200]  # a bright sky band so that the paths pick up energy
    return env


def _lightless_scene(pt):
    s = pt.Scene()
    grey = s.add_material([0.7, 0.7, 0.7], [0, 0, 0], [0, 0, 0])
    rough = s.add_material([0.6, 0.5, 0.4], [0, 0, 0], [0, 0, 0], roughness=0.15, mat_type=pt.MAT_OREN_NAYAR)
    gold = s.add_material([0, 0, 0], [0.944, 0.776, 0.373], [0, 0, 0], roughness=0.2, mat_type=pt.MAT_SPECULAR_METAL)
    mirror = s.add_material([0, 0, 0], [0.9, 0.9, 0.9], [0, 0, 0], roughness=0.01, mat_type=pt.MAT_SPECULAR_METAL)

    def quad(a, b, c, d, mat):
        a, b, c, d = (np.array(p, np.float32) for p in (a, b, c, d))
        n = np.cross(b - a, c - a)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        s.add_triangles(np.stack([np.concatenate([a, b, c]), np.concatenate([a, c, d])]), np.tile(n, (2, 3)), [mat, mat])

    quad([-3, -1, -3], [3, -1, -3], [3, -1, -9], [-3, -1, -9], grey)              # floor
    quad([-1.5, -1, -7], [0, -1, -7.5], [0, 1, -7.5], [-1.5, 1, -7], gold)        # two standing slabs
    quad([0.3, -1, -6.5], [1.8, -1, -6.0], [1.8, 0.8, -6.0], [0.3, 0.8, -6.5], rough)
    s.add_sphere([0.0, -0.4, -5.0], 0.6, mirror)
    # emits red only: FindEmissiveSpheres wants ALL Le components > 1e-4 (src/scene/sphere.cpp:11-24), so this emitter is in
    # no light list -- the scene stays "light-less" for NEE and its emission arrives through fc:889-896 / fc:1071-1077 alone
    red_glow = s.add_material([0, 0, 0], [0, 0, 0], [6.0, 0.0, 0.0], mat_type=pt.MAT_LIGHT)
    s.add_sphere([-1.6, 0.2, -5.5], 0.35, red_glow)
    s.finalize()
    assert s.arrays()["light_tris"].size == 0 and s.arrays()["light_spheres"].size == 0
    return s


@pytest.mark.parametrize("bounces", [0, 1, 5])
def test_scene_without_lights(pt, ol, bounces):
    scene, env = _lightless_scene(pt), _env()
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 4)
    c = pt.PathTracer(96, 64, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
    try:
        c.upload_scene(scene, env_rgb8=env)
        c.set_camera(cam)
        c.render(0, seeds, bounces)
        img = c.read_framebuffer()
        cnt = c.counters()
    finally:
        c.close()
    ref, ocnt = ol.render(ol.scene_from_desc(scene.desc(), env_rgb8=env), cam, 96, 64, seeds, bounces=bounces)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert cnt["rays_shadow"] == ocnt["rays_shadow"] == 0          # no light, no next-event estimation
    assert cnt["rays_bounce"] == ocnt["rays_bounce"]
    assert img[..., :3].max() > 0.0


def test_estimators_without_lights(pt, ol):
    scene, env = _lightless_scene(pt), _env()
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 3)
    for est, oflag in ((pt.PT_ESTIMATOR_NEE, 2), (pt.PT_ESTIMATOR_BRDF, 4)):
        c = pt.PathTracer(64, 64, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
        try:
            c.upload_scene(scene, env_rgb8=env)
            c.set_camera(cam)
            c.set_estimator(est)
            c.render(0, seeds, 4)
            img = c.read_framebuffer()
        finally:
            c.close()
        ref, _ = ol.render(ol.scene_from_desc(scene.desc(), env_rgb8=env), cam, 64, 64, seeds, bounces=4, flags=oflag)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), est


def test_empty_inputs_are_no_ops(pt):
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(pt.Scene.cornell())
        c.set_camera(pt.default_camera())
        prim, th, kind, uv = c.trace_closest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
        assert prim.size == 0 and kind.size == 0
        c.render(0, np.zeros(0, np.uint32), 5)                         # zero frames
        c.sync()
        assert not c.read_framebuffer().any()
        assert c.counters()["paths"] == 0
    finally:
        c.close()


def test_smallest_image_and_single_triangle(pt, ol):
    s = pt.Scene()
    m = s.add_material([0.5, 0.5, 0.5], [0, 0, 0], [3, 3, 3], mat_type=pt.MAT_LIGHT)
    v = np.array([[-1, -1, -5, 1, -1, -5, 0, 1, -5]], np.float32)
    n = np.array([[0, 0, 1, 0, 0, 1, 0, 0, 1]], np.float32)
    s.add_triangles(v, n, [m])
    s.finalize()                                                        # one triangle: the root is a leaf
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 2)
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(s)
        c.set_camera(cam)
        c.render(0, seeds, 5)
        img = c.read_framebuffer()
    finally:
        c.close()
    ref, _ = ol.render(ol.scene_from_desc(s.desc()), cam, 8, 8, seeds, bounces=5)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_contexts_release_their_memory(pt):
    """Twelve 1080p contexts in a row (23 GB of queues each at the default batch) and a scene re-upload loop:
    device memory must come back after pt_destroy / a new pt_upload_scene."""
    import torch
    torch.cuda.init()
    free0, total = torch.cuda.mem_get_info(0)
    scene = pt.Scene.soup(20000, seed=2)
    for _ in range(12):
        c = pt.PathTracer(1920, 1080)
        c.upload_scene(scene)
        c.set_camera(pt.default_camera())
        c.render(0, pt.frame_seeds(0, 1), 2)
        c.sync()
        c.close()
    c = pt.PathTracer(64, 64)
    try:
        for _ in range(50):
            c.upload_scene(scene)          # each upload frees the previous scene's buffers
        c.build_bvh_device(scene.arrays()["tris"], 2)
    finally:
        c.close()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 512 * 1024 * 1024, f"{(free0 - free1) / 2**20:.0f} MiB not returned"


def test_main_equivalent_driver(pt, tmp_path):
    """tools/pt355_main.cpp -- the reference's main() with the GL seam replaced -- renders frame by frame through
    pt_render_frame; its accumulator must equal the batched pt_render of the same frames bit for bit."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pt.__file__), "pt355_main")
    ppm, pfm = str(tmp_path / "m.ppm"), str(tmp_path / "m.pfm")
    out = subprocess.run([exe, "5", "256", "144", ppm, pfm], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Calculated BVH for scene" in out.stdout
    got = pt.read_pfm(pfm, 256, 144)
    c = pt.PathTracer(256, 144)
    try:
        c.upload_scene(pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175, spheres=True))
        c.set_camera(pt.default_camera())
        c.render(0, pt.frame_seeds(0, 5), 5)
        ref = c.read_framebuffer()
    finally:
        c.close()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert os.path.getsize(ppm) > 256 * 144 * 3


def test_main_equivalent_driver_with_a_gltf_model(pt, ol, tmp_path):
    """The same driver given a model file follows src/main.cpp:18-55 to the letter (LoadGLTF, Translate / Rotate / Scale /
    ApplyModelMatrixToTris, the four gold spheres): its image equals the Python view's (Scene.from_gltf) and the oracle's."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pt.__file__), "pt355_main")
    glb = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gltf", "cornell_lit.glb")
    ppm, pfm = str(tmp_path / "g.ppm"), str(tmp_path / "g.pfm")
    out = subprocess.run([exe, "4", "128", "96", ppm, pfm, glb], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    got = pt.read_pfm(pfm, 128, 96)
    sc = pt.Scene.from_gltf(glb)
    c = pt.PathTracer(128, 96)
    try:
        c.upload_scene(sc, tex_rgb8=sc.textures, tex_sampler=sc.tex_sampler)
        c.set_camera(pt.default_camera())
        c.render(0, pt.frame_seeds(0, 4), 5)
        mine = c.read_framebuffer()
    finally:
        c.close()
    assert np.array_equal(got.view(np.uint32), mine.view(np.uint32))
    ref, _ = ol.render(ol.scene_from_desc(sc.desc()), pt.default_camera(), 128, 96, pt.frame_seeds(0, 4), bounces=5)
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))
    assert mine[..., :3].max() > 0.1    # the emissive-strength ceiling panel lights the box


def test_retired_dead_paths_differ_only_where_the_reference_yields_nan(pt, ol):
    """retiring zero-throughput paths (the default) is 'same image' only while a dead path's later bounces stay
    finite.  A zero-area emissive triangle makes pdf_light_area infinite, the balance heuristic inf/inf = NaN, and the
    shader's color += throughput * direct turns 0 * NaN into NaN (framebuffer.comp:984-1006).  With
    PT_FLAG_TRACE_DEAD_PATHS the kernels reproduce those NaN pixels bit for bit; without it they stay finite there and
    equal the reference everywhere else."""
    s = pt.Scene()
    black = s.add_material([0, 0, 0], [0, 0, 0], [0, 0, 0])                       # Lambert, albedo 0: every path that lands here is dead
    grey = s.add_material([0.7, 0.7, 0.7], [0, 0, 0], [0, 0, 0])
    light = s.add_material([0, 0, 0], [0, 0, 0], [9, 9, 9], mat_type=pt.MAT_LIGHT)

    def quad(a, b, c, d, mat):
        a, b, c, d = (np.array(p, np.float32) for p in (a, b, c, d))
        n = np.cross(b - a, c - a)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        s.add_triangles(np.stack([np.concatenate([a, b, c]), np.concatenate([a, c, d])]), np.tile(n, (2, 3)), [mat, mat])
    quad([-3, -1, -3], [3, -1, -3], [3, -1, -9], [-3, -1, -9], black)             # floor
    quad([-3, -1, -9], [3, -1, -9], [3, 3, -9], [-3, 3, -9], grey)                # back wall
    quad([-1, 2.9, -5], [-1, 2.9, -7], [1, 2.9, -7], [1, 2.9, -5], light)         # a real light, facing down
    p = np.array([0.5, 1.0, -6.0] * 3, np.float32)                                # zero-area emissive triangle
    s.add_triangles(p[None, :], np.tile(np.array([0, 1, 0], np.float32), (1, 3)), [light])
    s.finalize()
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 4)
    ref, _ = ol.render(ol.scene_from_desc(s.desc()), cam, 96, 64, seeds, bounces=5)
    imgs = {}
    for flags in (pt.PT_FLAG_TRACE_DEAD_PATHS, 0):
        c = pt.PathTracer(96, 64, flags=flags)
        try:
            c.upload_scene(s)
            c.set_camera(cam)
            c.render(0, seeds, 5)
            imgs[flags] = c.read_framebuffer()
        finally:
            c.close()
    nan_ref = np.isnan(ref[..., :3]).any(axis=2)
    assert nan_ref.sum() > 0                                                       # the scene does provoke the case
    assert np.array_equal(imgs[pt.PT_FLAG_TRACE_DEAD_PATHS].view(np.uint32), ref.view(np.uint32))
    retired = imgs[0]
    same = (retired.view(np.uint32) == ref.view(np.uint32)).all(axis=2)
    assert same[~nan_ref].all()                                                    # identical wherever the reference is finite
    assert (~same).sum() > 0 and np.isfinite(retired[~same]).all()                 # and finite where only a dead path made it NaN


def test_upload_rejects_misshaped_textures(pt):
    c = pt.PathTracer(8, 8)
    try:
        with pytest.raises(pt.PtError, match="tex_rgb8"):
            c.upload_scene(pt.Scene.cornell(), tex_rgb8=np.zeros((2, 256, 256, 3), np.uint8))
        with pytest.raises(pt.PtError, match="env_rgb8"):
            c.upload_scene(pt.Scene.cornell(), env_rgb8=np.zeros((16, 32), np.uint8))
    finally:
        c.close()
