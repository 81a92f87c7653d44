"""Three more pins of the oracle by sources that do not share its text:

  (a) the pinhole camera of render_function (framebuffer.comp:1088-1107) against its geometry in float64: the image plane
      2 units ahead of the eye and 2 units high, i.e. a half field of view of atan(1/2) = 26.565 deg vertically and
      atan(W/H / 2) horizontally, pixel (0, 0) bottom-left, jitter = two rand() draws minus 0.5 (x first);
  (b) the running mean of main() (fc:1124-1129): after n frames the image is the arithmetic mean of the n samples;
  (c) the narrow members of HitData (fc:214-222): mat_index is uint8 (a material index >= 256 shades with index & 255),
      object_index is uint16 (a hit on light triangle i >= 65536 takes its MIS area from triangle i & 65535: fc:299-300, :1050).

No GPU except the last test, which checks the HIP path on the same two scenes."""
import ctypes

import numpy as np
import pytest


# ------------------------------------------------------------------ (a) pinhole rays
def _pcg_stream(x, y, seed, n):
    st = (19 * x + 47 * y + 101 * seed + 131) & 0xffffffff
    out = []
    for _ in range(n):
        old, st = st, (st * 747796405 + 2891336453) & 0xffffffff
        w = (((old >> ((old >> 28) + 4)) ^ old) * 277803737) & 0xffffffff
        out.append(((w >> 22) ^ w) & 0xffffffff)
    return out, st


def _rand(h):
    return min(max(float(np.float32(h) * np.float32(2.0 ** -32)), 1e-4), 1.0 - 1e-4)


def test_pinhole_rays_match_the_camera_geometry(pt, ol):
    rng = np.random.default_rng(11)
    recs, want, states = [], [], []
    for case in range(400):
        W, H = [(1920, 1080), (64, 64), (8, 24), (1280, 720), (3840, 2160)][case % 5]
        if case < 200:
            o, f, r = np.zeros(3), np.array([0.0, 0.0, -1.0]), np.array([1.0, 0.0, 0.0])   # src/main.cpp:67
        else:
            o = rng.uniform(-3, 3, 3)
            f = rng.normal(size=3); f /= np.linalg.norm(f)
            r = np.cross(f, rng.normal(size=3)); r /= np.linalg.norm(r)
        px, py = [(0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1), (W // 2, H // 2)][case % 5] if case % 2 else (int(rng.integers(0, W)), int(rng.integers(0, H)))
        seed = int(rng.integers(0, 2 ** 32))
        cam = np.zeros(12, np.float32)
        cam[0:3], cam[4:7], cam[8:11] = o, f, r
        o32, f32, r32 = cam[0:3].astype(np.float64), cam[4:7].astype(np.float64), cam[8:11].astype(np.float64)
        (h1, h2), st = _pcg_stream(px, py, seed, 2)
        jx, jy = _rand(h1) - 0.5, _rand(h2) - 0.5
        up = np.cross(r32, f32); up /= np.linalg.norm(up)
        # the plane is 2 ahead, 2 high and 2 W/H wide; pixel centres sit on integer coordinates, (0, 0) at the bottom-left corner
        x_plane = ((px + jx) / W - 0.5) * 2.0 * W / H
        y_plane = ((py + jy) / H - 0.5) * 2.0
        d = 2.0 * f32 + x_plane * r32 + y_plane * up
        want.append(np.concatenate([o32, d / np.linalg.norm(d)]))
        states.append(st)
        recs.append(np.concatenate([cam, np.array([W, H, px, py], np.float32), np.array([seed], np.uint32).view(np.float32)]))
    got = ol.probe("PRIMARY_RAY", np.array(recs, np.float32))
    want = np.array(want)
    assert np.array_equal(got[:, :3], want[:, :3].astype(np.float32))            # the ray starts at the eye, exactly
    assert np.abs(got[:, 3:6].astype(np.float64) - want[:, 3:]).max() < 4e-7     # fp32 evaluation of the same geometry
    assert np.array_equal(got[:, 6].view(np.uint32), np.array(states, np.uint32))  # exactly two draws were taken
    # the field of view, from the four image corners of the default camera: tan = 1/2 vertically, W/H / 2 horizontally
    # (to within the half-pixel the jitter moves a corner pixel's ray)
    for i in range(1, 200, 2):
        W, H, px, py = recs[i][12:16]
        if (px, py) == (W - 1, H - 1):
            d = got[i, 3:6].astype(np.float64)
            assert abs(d[1] / -d[2] - 0.5) < 1.5 / H and abs(d[0] / -d[2] - 0.5 * W / H) < 1.5 / H   # the last pixel's centre is 1 pixel inside
            assert abs(np.degrees(np.arctan(d[1] / -d[2])) - 26.565) < np.degrees(1.5 / H)
        if (px, py) == (0, 0):
            assert got[i, 3] < 0 and got[i, 4] < 0                                  # pixel (0, 0) is bottom-left
        if (px, py) == (W // 2, H // 2):
            assert np.abs(got[i, 3:5]).max() < 1.5 / min(W, H) and got[i, 5] < -0.999   # the image centre looks along `forward`


# ------------------------------------------------------------------ (b) running mean
@pytest.mark.parametrize("n", [1, 2, 3, 7, 64])
def test_running_mean_is_the_arithmetic_mean(pt, ol, n):
    sc = pt.Scene.cornell(spheres=True)
    osc, cam = ol.scene_from_desc(sc.desc()), pt.default_camera()
    seeds = pt.frame_seeds(0, n)
    img, _ = ol.render(osc, cam, 24, 16, seeds, bounces=3)
    samples = [ol.render(osc, cam, 24, 16, seeds[f: f + 1], first_frame=0, bounces=3)[0].astype(np.float64) for f in range(n)]
    mean = np.mean(samples, axis=0)
    err = np.abs(img[..., :3] - mean[..., :3]) / np.maximum(mean[..., :3], 1e-3)
    assert err.max() < 1e-6 * max(1, n // 8), err.max()   # weights (f/(f+1), 1/(f+1)) telescope to 1/n each
    assert (img[..., 3] == 1.0).all()


# ------------------------------------------------------------------ (c) u8 material index, u16 object index
def _quad(s, a, b, c, d, mat):
    a, b, c, d = (np.array(p, np.float32) for p in (a, b, c, d))
    n = np.cross(b - a, c - a)
    n = (n / np.linalg.norm(n)).astype(np.float32)
    s.add_triangles(np.stack([np.concatenate([a, b, c]), np.concatenate([a, c, d])]), np.tile(n, (2, 3)), [mat, mat])


def _box(s, mats, light):
    x0, x1, y0, y1, z0, z1 = -2.0, 2.0, -2.0, 2.0, -8.0, -4.0
    _quad(s, [x0, y0, z1], [x1, y0, z1], [x1, y0, z0], [x0, y0, z0], mats[0])
    _quad(s, [x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], mats[1])
    _quad(s, [x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1], mats[2])
    _quad(s, [x0, y0, z1], [x0, y0, z0], [x0, y1, z0], [x0, y1, z1], mats[3])
    _quad(s, [x1, y0, z0], [x1, y0, z1], [x1, y1, z1], [x1, y1, z0], mats[4])
    _quad(s, [-0.7, 1.99, -6.7], [0.7, 1.99, -6.7], [0.7, 1.99, -5.3], [-0.7, 1.99, -5.3], light)


def test_material_index_is_truncated_to_8_bits(pt, ol):
    def build(wall_mats):
        lm = 7  # the light's material: a low index, so that the shader and FindEmissiveTris agree on it
        s2 = pt.Scene()
        for m in range(300):
            le = [17, 12, 4] if m == lm else [0, 0, 0]
            ty = pt.MAT_LIGHT if m == lm else pt.MAT_LAMBERTIAN
            s2.add_material([0.2 + 0.6 * ((m * 37) % 11) / 10.0, 0.2 + 0.6 * ((m * 53) % 7) / 6.0, 0.2 + 0.6 * ((m * 29) % 5) / 4.0], [0, 0, 0], le, mat_type=ty)
        _box(s2, wall_mats, lm)
        s2.finalize()
        return s2
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 2)
    high = [256 + k for k in (10, 11, 12, 13, 14)]         # what the scene says
    low = [k for k in (10, 11, 12, 13, 14)]                # what uint8_t(tri.data4.w) makes of it (fc:299)
    s_high, s_low = build(high), build(low)   # kept alive: a desc points into its scene
    a, _ = ol.render(ol.scene_from_desc(s_high.desc()), cam, 48, 48, seeds)
    b, _ = ol.render(ol.scene_from_desc(s_low.desc()), cam, 48, 48, seeds)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # and the colours of materials 266..270 themselves are different ones: without the truncation the image would be another
    sa = s_high.arrays()["materials"].view(np.float32)
    assert not np.allclose(sa[266:271, :3], sa[10:15, :3])
    assert (a[..., :3] > 0).mean() > 0.5


def _scene_with_a_light_past_index_65535(pt):
    """66,000 small triangles far to the LEFT of a closed-in box (no ray from inside the box, or leaving through its open front,
    can reach them), then the box: the builder's leaf order follows x, so the box and its light get sorted indices >= 66,000."""
    s = pt.Scene()
    white = s.add_material([0.73, 0.73, 0.73], [0, 0, 0], [0, 0, 0])
    light = s.add_material([0, 0, 0], [0, 0, 0], [17, 12, 4], mat_type=pt.MAT_LIGHT)
    rng = np.random.default_rng(3)
    n = 66000
    c = np.stack([rng.uniform(-200, -100, n), rng.uniform(-2, 2, n), rng.uniform(-8, -6, n)], 1)[:, None, :]
    v = (c + rng.uniform(-0.05, 0.05, (n, 3, 3))).astype(np.float32)
    nn = np.tile(np.array([0, 0, 1], np.float32), (n, 3))
    s.add_triangles(v.reshape(n, 9), nn, np.full(n, white, np.uint32))
    _box(s, [white] * 5, light)
    s.finalize()
    return s


def _desc_with_triangles(pt, scene, tris):
    d = scene.desc()
    d.tris = tris.ctypes.data_as(ctypes.c_void_p).value
    return d


def _shrunk(tris_u32, j):
    """triangle j scaled by 1/2 about its centroid: a quarter of the area, still inside its leaf's box"""
    t = tris_u32.copy()
    f = t.view(np.float32)
    v = np.stack([f[j, 0:3], f[j, 4:7], f[j, 8:11]]).astype(np.float64)
    c = v.mean(0)
    v = (c + 0.5 * (v - c)).astype(np.float32)
    f[j, 0:3], f[j, 4:7], f[j, 8:11] = v
    return t


def _u16_case(pt):
    scene = _scene_with_a_light_past_index_65535(pt)
    a = scene.arrays()
    lights = a["light_tris"]
    assert len(lights) == 2 and lights.min() >= 65536, lights
    j = int(lights[0]) - 65536          # the triangle uint16_t(tri_index) points at instead (fc:300)
    assert a["tris"].view(np.float32)[j, 0] < -50   # one of the hidden ones
    assert int(lights[1]) - 65536 != j + 7
    return scene, a["tris"], j


def test_object_index_is_truncated_to_16_bits_and_feeds_the_light_area(pt, ol):
    scene, tris, j = _u16_case(pt)
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 3)
    base, _ = ol.render(ol.scene_from_desc(scene.desc()), cam, 48, 48, seeds)
    t_alias, t_other = _shrunk(tris, j), _shrunk(tris, j + 7)   # j + 7: a hidden triangle no light index wraps to
    alias, _ = ol.render(ol.scene_from_desc(_desc_with_triangles(pt, scene, t_alias)), cam, 48, 48, seeds)
    other, _ = ol.render(ol.scene_from_desc(_desc_with_triangles(pt, scene, t_other)), cam, 48, 48, seeds)
    # nobody can SEE triangle j or j + 7; the only way j reaches a pixel is as area_triangle(triangles[object_index]) in the MIS
    # weight of a path that hit the light (fc:1048-1063)
    assert np.array_equal(base.view(np.uint32), other.view(np.uint32))
    changed = (base.view(np.uint32) != alias.view(np.uint32)).any(axis=2)
    assert changed.sum() > 50, changed.sum()


@pytest.mark.gpu
def test_gpu_reproduces_the_u8_and_u16_truncations(pt, ol):
    scene, tris, j = _u16_case(pt)
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 3)
    t_alias = _shrunk(tris, j)
    for d in (scene.desc(), _desc_with_triangles(pt, scene, t_alias)):
        c = pt.PathTracer(48, 48, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
        try:
            c.upload_scene(d)
            c.set_camera(cam)
            c.render(0, seeds, 5)
            img = c.read_framebuffer()
        finally:
            c.close()
        ref, _ = ol.render(ol.scene_from_desc(d), cam, 48, 48, seeds)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
