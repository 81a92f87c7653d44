"""Pins of the INTEGRATOR half of the oracle by sources that do not share its text:

  (a) the reference's own CPU twins of the shader's sampling helpers (src/math/math.cpp:132-194), run here through
      oracle/_ref and committed as tests/golden/ref_math.npz;
  (b) closest hits found by the reference's vendored BVH library (SingleRayTraverser + ClosestPrimitiveIntersector),
      same file;
  (c) closed forms, quadrature and histograms written here in numpy from the published formulas: GGX normalisation,
      the cosine lobe's density, the GGX sampling density, Oren-Nayar's sigma -> 0 limit, the irradiance of a parallel
      rectangle, the geometric series of a closed furnace, agreement of the three estimators.

No GPU; everything through oracle/liboracle.so's probe hooks."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def G(golden):
    return golden("ref_math.npz")


def _ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


# ------------------------------------------------------------------ (a) the reference's CPU twins
def test_map_to_triangle_equals_the_reference_twin(ol, G):
    # framebuffer.comp:707-723 == src/math/math.cpp:159-175: same expression tree -> same bits
    rec = np.concatenate([G["uv"], G["tri_verts"]], 1)
    got = ol.probe("MAP_TO_TRIANGLE", rec)
    assert (G["uv"].sum(1) > 1).sum() > 1000  # the reflected half of the parallelogram is exercised
    assert np.array_equal(got.view(np.uint32), G["map_to_triangle"].view(np.uint32))


def test_tnb_equals_the_reference_twin(ol, G):
    # framebuffer.comp:141-152 (transposed at :907) == src/math/math.cpp:178-194, rows t, n, bt
    got = ol.probe("TNB_ROWS", G["tnb_normals"])
    assert (G["tnb_normals"][:, 2] < 0).sum() > 1000  # both signs of the branch
    assert np.array_equal(got.view(np.uint32), G["tnb_rows"].view(np.uint32))


def test_map_to_unit_sphere_equals_the_reference_twin_under_libm(ol, G):
    # framebuffer.comp:154-170 == src/math/math.cpp:132-151; the twin calls glibc's sinf/cosf, which is what
    # ORACLE_FLAG_LIBM selects: identical bits.  The contract's polynomial forms stay within 1e-6 of it.
    got = ol.probe("MAP_TO_UNIT_SPHERE", G["uv"], flags=ol.FLAG_LIBM)
    assert np.array_equal(got.view(np.uint32), G["map_to_unit_sphere"].view(np.uint32))
    spec = ol.probe("MAP_TO_UNIT_SPHERE", G["uv"])
    assert np.abs(spec.astype(np.float64) - G["map_to_unit_sphere"]).max() < 1e-6


def test_cosine_hemisphere_matches_the_reference_twin(ol, G):
    # framebuffer.comp:172-175 == src/math/math.cpp:153-157.  glm::normalize multiplies by inversesqrt, the oracle divides
    # by sqrt (the GLSL spec leaves normalize's form open): a few ulps where the sum is not nearly cancelled.
    rec = np.concatenate([G["uv"], G["normals"]], 1)
    got = ol.probe("COSINE_HEMISPHERE", rec, flags=ol.FLAG_LIBM)
    want = G["cosine_hemisphere"]
    assert np.abs(got.astype(np.float64) - want).max() < 5e-7
    assert np.percentile(_ulps(got, want), 99) <= 4


def test_oct_decode_inverts_the_reference_encoder(pt, ol, golden):
    # oct encode is the REFERENCE'S (src/math/math.cpp:196-211, pinned bit for bit in ref_host.npz); decoding what it
    # produced (framebuffer.comp:203-210) must give the normal back to fp16 accuracy
    H = golden("ref_host.npz")
    n = H["oct_in"]
    n = n / np.linalg.norm(n, axis=1, keepdims=True)
    enc = H["oct_out"].astype(np.float32)
    packed = np.array([pt.lib().pth_pack_half2x16(float(e[0]), float(e[1])) for e in enc], np.uint32)
    dec = ol.probe("OCT_DECODE", packed.view(np.float32))
    assert np.abs(dec - n).max() < 4e-3
    assert np.abs(np.linalg.norm(dec, axis=1) - 1).max() < 2e-3


# ------------------------------------------------------------------ (b) the vendored BVH library's own closest hit
@pytest.mark.parametrize("name", ["cornell", "soup3000", "bunny12"])
def test_closest_hit_against_the_reference_bvh_library(pt, ol, G, name):
    sc = {"cornell": lambda: pt.Scene.cornell(), "soup3000": lambda: pt.Scene.soup(3000, seed=5),
          "bunny12": lambda: pt.Scene.bunny(freq=12, target_tris=0)}[name]()
    assert sc.hash() == int(G[f"hit_{name}_scene_hash"][0])
    ro, rd = G[f"hit_{name}_ro"], G[f"hit_{name}_rd"]
    rprim, rt = G[f"hit_{name}_prim"], G[f"hit_{name}_t"]
    prim, th, kind, _ = ol.trace_closest(ol.scene_from_desc(sc.desc()), ro, rd)
    t = th.view(np.float16).astype(np.float32)
    rhit, ohit = rprim != 0xffffffff, kind == 1
    # hit / miss: may differ only for grazing hits the shader's |det| < 1e-4 cut-off or its u,v edge rules drop
    both = rhit & ohit
    assert both.sum() > 0.5 * ro.shape[0]
    assert (rhit != ohit).mean() < 0.01
    # distance: the shader keeps fp16(t) of an fp32 Moeller-Trumbore; the library keeps its own fp32 t
    rel = np.abs(t[both] - rt[both]) / np.maximum(rt[both], 1e-3)
    assert np.percentile(rel, 99) < 2e-3 and rel.max() < 0.05
    # primitive: identical wherever the two t agree to fp16 accuracy and no other triangle is a near-tie
    same = prim[both] == rprim[both]
    assert same.mean() > 0.97
    # every disagreement on the primitive is a near-tie in distance (coplanar / shared-edge triangles), not a different surface
    assert (rel[~same] < 5e-3).all()


# ------------------------------------------------------------------ (c) closed forms written from the published formulas
def test_ggx_ndf_is_normalised(ol):
    # Walter et al. 2007: the integral of D(m) (m . n) over the hemisphere is 1, for every alpha
    th = (np.arange(20000) + 0.5) / 20000 * (np.pi / 2)
    for alpha in (0.05, 0.2, 0.5, 1.0):
        wm = np.stack([np.sin(th), np.cos(th), np.zeros_like(th)], 1)
        D = ol.probe("NDF", np.concatenate([wm, np.full((th.size, 1), alpha)], 1))[:, 0].astype(np.float64)
        integral = (D * np.cos(th) * np.sin(th) * 2 * np.pi).sum() * (np.pi / 2 / th.size)
        assert abs(integral - 1.0) < 2e-3, (alpha, integral)
        # and it is the Trowbridge-Reitz form: D = a^2 / (pi ((a^2 - 1) cos^2 + 1)^2)
        a2 = alpha * alpha
        want = a2 / (np.pi * ((a2 - 1.0) * np.cos(th) ** 2 + 1.0) ** 2)
        assert np.abs(D - want).max() / want.max() < 2e-4  # fp32 inputs near the peak of a narrow lobe


def _pick(ol, n, mat_type, rough, wo, seed):
    st = (np.arange(n, dtype=np.uint64) * 2654435761 + seed).astype(np.uint32)
    rec = np.zeros((n, 6), np.float32)
    rec[:, 0] = st.view(np.float32)
    rec[:, 1:4] = wo
    rec[:, 4], rec[:, 5] = mat_type, rough
    o = ol.probe("PICK_WI", rec)
    return o[:, 0:3], o[:, 3:6], o[:, 6], o[:, 7]


def test_cosine_lobe_density(ol):
    # pick_wi for diffuse materials (framebuffer.comp:565-573): wi ~ cos(theta) / pi about (0,1,0); pdf and cos returned consistently
    n = 400000
    wi, wm, c, pdf = _pick(ol, n, 0.0, 0.0, [0.3, 0.9, 0.1], 7)
    assert np.abs(np.linalg.norm(wi, axis=1) - 1).max() < 1e-5
    assert np.array_equal(c, wi[:, 1]) or np.abs(c - wi[:, 1]).max() < 1e-6
    assert np.abs(pdf - c / np.pi).max() < 1e-6
    assert (wi[:, 1] > 0).all()
    # cos(theta)^2 is uniform on [0,1] under the cosine density; azimuth is uniform: chi-square on a 10 x 8 grid
    u = np.clip((wi[:, 1].astype(np.float64) ** 2 * 10).astype(int), 0, 9)
    phi = np.arctan2(wi[:, 2], wi[:, 0]) + np.pi
    v = np.clip((phi / (2 * np.pi) * 8).astype(int), 0, 7)
    counts = np.bincount(u * 8 + v, minlength=80).astype(np.float64)
    chi2 = ((counts - n / 80) ** 2 / (n / 80)).sum()
    assert chi2 < 140, chi2  # 79 degrees of freedom: P(chi2 > 140) < 3e-5
    # E[cos] under cos/pi is 2/3
    assert abs(wi[:, 1].mean() - 2 / 3) < 2e-3


@pytest.mark.parametrize("alpha", [0.1, 0.3, 0.7])
def test_ggx_sampling_density(ol, alpha):
    # importance_sample_trowbridge_reitz (framebuffer.comp:479-499): wm ~ D(m) (m . n);  CDF of cos^2(theta_m):
    # P(cos^2 >= c) = e(c) with cos^2 = (1 - e) / ((a^2 - 1) e + 1)  <=>  e = (1 - c) / ((a^2 - 1) c + 1)
    n = 300000
    wo = np.array([0.4, 0.8, 0.2]) / np.linalg.norm([0.4, 0.8, 0.2])
    wi, wm, c, pdf = _pick(ol, n, 2.0, alpha, wo, 11)
    assert (c == 1.0).all() and (pdf == 1.0).all()                 # reference quirk 9: cos = pdf = 1 for GGX samples
    assert np.abs(np.linalg.norm(wm, axis=1) - 1).max() < 1e-5
    # density check by quadrature of D(m) cos sin 2 pi over theta bins (D written here, not taken from the oracle)
    a2 = alpha * alpha
    edges = np.linspace(0, np.pi / 2, 33)
    th = np.arccos(np.clip(wm[:, 1].astype(np.float64), -1, 1))
    counts = np.histogram(th, edges)[0].astype(np.float64)
    fine = np.linspace(0, np.pi / 2, 32 * 400 + 1)
    mid = 0.5 * (fine[1:] + fine[:-1])
    dens = a2 / (np.pi * ((a2 - 1.0) * np.cos(mid) ** 2 + 1.0) ** 2) * np.cos(mid) * np.sin(mid) * 2 * np.pi
    expect = dens.reshape(32, 400).sum(1) * (fine[1] - fine[0]) * n
    keep = expect > 20
    chi2 = ((counts[keep] - expect[keep]) ** 2 / expect[keep]).sum()
    assert chi2 < 90, (alpha, chi2)
    # wi is the mirror direction of wo about wm
    refl = 2.0 * (wm * wo).sum(1, keepdims=True) * wm - wo
    refl /= np.linalg.norm(refl, axis=1, keepdims=True)
    assert np.abs(wi - refl).max() < 2e-5


def _brdf(ol, wo, wm, wi, diffuse, f0, mat_type, rough, nee):
    n = len(wo)
    rec = np.zeros((n, 18), np.float32)
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9] = wo, wm, wi
    rec[:, 9:12], rec[:, 12:15] = diffuse, f0
    rec[:, 15], rec[:, 16], rec[:, 17] = mat_type, rough, 1.0 if nee else 0.0
    return ol.probe("CALC_BRDF", rec)


def _hemi(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v[:, 1] = np.abs(v[:, 1]) + 0.05
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_brdf_closed_forms(ol):
    wo, wi = _hemi(2000, 1), _hemi(2000, 2)
    up = np.tile(np.array([0, 1, 0], np.float32), (2000, 1))
    alb = np.array([0.73, 0.5, 0.25], np.float32)
    # Lambert: albedo / pi; light material: 0; unknown type: (1,1,0) (framebuffer.comp:593)
    assert np.allclose(_brdf(ol, wo, up, wi, alb, 0, 0.0, 0.0, False), alb / np.pi, rtol=1e-6)
    assert (_brdf(ol, wo, up, wi, alb, 0, -1.0, 0.0, False) == 0).all()
    assert (_brdf(ol, wo, up, wi, alb, 0, 5.0, 0.0, False) == np.array([1, 1, 0], np.float32)).all()
    # Oren-Nayar (framebuffer.comp:458-472): sigma -> 0 is Lambert exactly (A = 1, B = 0) ...
    assert np.allclose(_brdf(ol, wo, up, wi, alb, 0, 1.0, 0.0, False), alb / np.pi, rtol=1e-6)
    # ... and for sigma > 0 the qualitative model's closed form, written here from Oren & Nayar 1994 eq. 30
    s2 = 0.35 ** 2
    A, B = 1 - 0.5 * s2 / (s2 + 0.33), 0.45 * s2 / (s2 + 0.09)
    ti, to = np.arccos(wi[:, 1].astype(np.float64)), np.arccos(wo[:, 1].astype(np.float64))
    # the shader's azimuth is atan(x, z): cos(phi_i - phi_o) from the projected directions
    pi_, po_ = np.arctan2(wi[:, 0], wi[:, 2]), np.arctan2(wo[:, 0], wo[:, 2])
    k = A + B * np.maximum(0, np.cos(pi_ - po_)) * np.sin(np.maximum(ti, to)) * np.tan(np.minimum(ti, to))
    want = alb[None, :] / np.pi * k[:, None]
    got = _brdf(ol, wo, up, wi, alb, 0, 1.0, 0.35, False)
    assert np.abs(got - want).max() / want.max() < 2e-5
    # GGX, NEE form (framebuffer.comp:540-549) = F G2 D / (4 |n.wi| |n.wo|) with Smith's separable-height Lambda
    f0 = np.array([0.944, 0.776, 0.373])
    for alpha in (0.1, 0.4):
        got = _brdf(ol, wo, up, wi, 0, f0, 2.0, alpha, True).astype(np.float64)
        h = (wo + wi).astype(np.float64)
        h /= np.linalg.norm(h, axis=1, keepdims=True)
        a2 = alpha * alpha
        D = a2 / (np.pi * ((a2 - 1) * h[:, 1] ** 2 + 1) ** 2)

        def lam(w):
            t2 = (1 - w[:, 1].astype(np.float64) ** 2) / w[:, 1].astype(np.float64) ** 2
            return -0.5 + 0.5 * np.sqrt(1 + a2 * t2)
        G2 = 1.0 / (1 + lam(wo) + lam(wi))
        x = 1 - np.clip((h * wi).sum(1), 0, 1)
        F = f0[None, :] + (1 - f0[None, :]) * (x ** 5)[:, None]
        want = F * (G2 * D / (4 * wi[:, 1] * wo[:, 1]))[:, None]
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-4, alpha


def test_inverse_of_the_tnb_matrix_is_its_transpose(ol, G):
    # framebuffer.comp:908 takes a true inverse() of an orthonormal matrix: it must equal the transpose to rounding
    rows = ol.probe("TNB_ROWS", G["tnb_normals"][:2000])
    inv = ol.probe("INVERSE3", rows).reshape(-1, 3, 3)
    assert np.abs(inv - rows.reshape(-1, 3, 3).transpose(0, 2, 1)).max() < 2e-5


# ---- radiance against closed forms: two parallel planes (cos_x = cos_y for every connection, so the reference's
# cos_y quirk at framebuffer.comp:1066 cancels; right-angled light triangles, so the |e1||e2|/2 area quirk is exact)
def _plane_scene(pt, rho, le, light_half=0.5, h=1.0, floor_half=50.0, light_type=None):
    s = pt.Scene()
    mf = s.add_material([rho] * 3, [0, 0, 0], [0, 0, 0], 0.0, -1, pt.MAT_LAMBERTIAN)
    ml = s.add_material([0, 0, 0], [0, 0, 0], [le] * 3, 0.0, -1, pt.MAT_LIGHT if light_type is None else light_type)

    def quad(y, half, up):
        a, b, c, d = [-half, y, -half], [half, y, -half], [half, y, half], [-half, y, half]
        # geometric normal = shading normal; v0 is the RIGHT-ANGLE vertex of each triangle, so that the reference's light area
        # sqrt(|e1|^2 |e2|^2) / 2 (framebuffer.comp:695-700) is the true area
        v = np.array([b + a + c, d + c + a], np.float32) if up else np.array([b + c + a, d + a + c], np.float32)
        n = np.tile(np.array([0, 1 if up else -1, 0], np.float32), (2, 3))
        return v, n
    v, n = quad(0.0, floor_half, True)
    s.add_triangles(v, n, [mf, mf])
    v, n = quad(h, light_half, False)
    s.add_triangles(v, n, [ml, ml])
    s.finalize()
    return s


def _form_factor_parallel_rect(a, b, h):
    # irradiance factor (1/pi) * integral cos cos / d^2 dA of a point under the CORNER of an a x b rectangle at height h
    A, B = a / h, b / h
    return (1 / (2 * np.pi)) * (A / np.sqrt(1 + A * A) * np.arctan(B / np.sqrt(1 + A * A)) + B / np.sqrt(1 + B * B) * np.arctan(A / np.sqrt(1 + B * B)))


def test_direct_lighting_matches_the_closed_form(pt, ol):
    rho, le, half, h = 0.6, 10.0, 0.5, 1.0
    sc = _plane_scene(pt, rho, le, half, h)
    osc = ol.scene_from_desc(sc.desc())
    # camera ray from just under the light plane straight down onto the floor point (x0, 0, z0)
    for x0, z0 in ((0.0, 0.0), (0.3, -0.2)):
        ro, rd = [x0 + 0.15, 0.9, z0 + 0.1], np.array([-0.15, -0.9, -0.1])
        rd = rd / np.linalg.norm(rd)
        # point under the rectangle's interior: sum of the four corner form factors
        ff = sum(_form_factor_parallel_rect(abs(sx * half - x0), abs(sz * half - z0), h) for sx in (-1, 1) for sz in (-1, 1))
        want = rho * le * ff   # L_o = (rho / pi) * E,  E = pi * Le * ff
        for flags, bounces, name in ((0, 1, "mis"), (ol.FLAG_EST_BRDF, 2, "brdf"), (ol.FLAG_EST_NEE, 2, "nee")):
            got = ol.estimate(osc, ro, rd, 200000, bounces=bounces, flags=flags, seed=3)
            assert np.allclose(got, got[0])
            # MC error ~0.3 %, fp16 hit distance / normal quantisation ~0.2 %
            assert abs(got[0] - want) / want < 0.015, (name, x0, z0, got[0], want)


def test_furnace_geometric_series_of_the_brdf_estimator(pt, ol):
    # estimator_path_tracing_BRDF (framebuffer.comp:647-693) adds throughput * Le at every hit and multiplies the
    # throughput by cos * (rho / pi) / (cos / pi) = rho: inside a closed box whose walls all carry (rho, Le) the result is
    # Le * (1 + rho + ... + rho^(b-1)) for b intersect() calls -- no Monte-Carlo error, only rounding.
    rho, le = 0.5, 2.0
    s = pt.Scene()
    m = s.add_material([rho] * 3, [0, 0, 0], [le] * 3, 0.0, -1, pt.MAT_LAMBERTIAN)
    c = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32)
    faces = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    v, n = [], []
    for f in faces:
        p = c[list(f)]
        nn = -np.sign(p.mean(0))  # inward
        v += [np.concatenate([p[0], p[1], p[2]]), np.concatenate([p[0], p[2], p[3]])]
        n += [np.tile(nn, 3)] * 2
    s.add_triangles(np.array(v, np.float32), np.array(n, np.float32), [m] * 12)
    s.finalize()
    osc = ol.scene_from_desc(s.desc())
    for b in (1, 2, 5):
        got = ol.estimate(osc, [0.1, 0.2, -0.1], [0.3, -0.5, 0.81], 2000, bounces=b, flags=ol.FLAG_EST_BRDF, seed=5)
        want = le * sum(rho ** k for k in range(b))
        assert abs(got[0] - want) / want < 1e-4, (b, got, want)


def test_sphere_hits_match_the_quadratic(pt, ol):
    # intersect_sphere + the sphere loop of intersect (framebuffer.comp:226-268, 439-453) against the textbook quadratic
    # in float64, on a scene whose triangles are out of reach of the rays (only spheres can be hit)
    s = pt.Scene()
    m = s.add_material([0.5] * 3, [0, 0, 0], [0, 0, 0], 0.0, -1, pt.MAT_LAMBERTIAN)
    far = np.array([[50, 50, 50, 51, 50, 50, 50, 51, 50], [60, 60, 60, 61, 60, 60, 60, 61, 60]], np.float32)
    s.add_triangles(far, np.tile(np.array([0, 0, 1], np.float32), (2, 3)), [m, m])
    centres = np.array([[0, 0, -5], [1.0, 0.5, -4], [-1.2, -0.3, -6], [0.2, 1.1, -3]], np.float64)
    radii = np.array([1.0, 0.5, 0.8, 0.3])
    for c, r in zip(centres, radii):
        s.add_sphere(c, float(r), m)
    s.finalize()
    rng = np.random.default_rng(5)
    n = 20000
    ro = rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)) * [0.4, 0.4, 0.2] + [0, 0, -1]
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    prim, th, kind, _ = ol.trace_closest(ol.scene_from_desc(s.desc()), ro, rd)
    t = th.view(np.float16).astype(np.float64)
    o, d = ro.astype(np.float64), rd.astype(np.float64)
    best_t, best_i = np.full(n, np.inf), np.full(n, -1)
    for i, (c, r) in enumerate(zip(centres, radii)):
        oc = o - c
        b = (oc * d).sum(1)
        disc = b * b - ((oc * oc).sum(1) - r * r)
        tt = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf)
        tt = np.where(tt > 1e-3, tt, np.inf)
        closer = tt < best_t
        best_t, best_i = np.where(closer, tt, best_t), np.where(closer, i, best_i)
    hit = np.isfinite(best_t)
    assert hit.sum() > 3000
    assert ((kind == 2) == hit).mean() > 0.999        # grazing rays may fall either way
    both = hit & (kind == 2)
    clear = both & (np.abs(t - best_t) < 2e-2)
    assert clear.sum() > 0.999 * both.sum()
    assert (np.abs(t[clear] - best_t[clear]) / best_t[clear]).max() < 1.5e-3   # fp16 t
    assert (prim[clear] == best_i[clear]).mean() > 0.999
