"""GPU BVH build (pt_build_bvh_device): the tree must be a valid input of everything that takes the
reference builder's output -- structure checked on the host, closest hits bit-exact against the oracle walking the SAME
tree, frames bit-identical, and the hits must agree with those found through the Sweep-SAH tree of the same triangles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    ro = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    ro[: n // 2] = 0.0
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    return ro, rd


def _check_tree(nodes_u32, tris_u32, max_leaf):
    """Every triangle in exactly one leaf, every box holds what is below it, siblings adjacent, depth <= 64."""
    nodes = nodes_u32.view(np.float32)
    tris = tris_u32.view(np.float32)
    n_nodes, n_tris = nodes.shape[0], tris.shape[0]
    assert n_nodes % 2 == 1 and n_nodes >= 3
    verts = tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(n_tris, 3, 3)
    tlo, thi = verts.min(1), verts.max(1)
    seen = np.zeros(n_tris, np.int32)
    claimed = np.zeros(n_nodes, np.int32)
    assert nodes[0, 3] == 1.0 and nodes[0, 7] == 0.0
    max_depth = 0
    stack = [(0, 1)]
    while stack:
        i, depth = stack.pop()
        max_depth = max(max_depth, depth)
        lo, hi = nodes[i, 0:3], nodes[i, 4:7]
        first, count = int(nodes[i, 3]), int(nodes[i, 7])
        if count > 0:
            assert count <= max(max_leaf, 1)
            assert first + count <= n_tris
            seen[first:first + count] += 1
            assert (tlo[first:first + count] >= lo).all() and (thi[first:first + count] <= hi).all()
        else:
            assert 1 <= first and first + 1 < n_nodes and first % 2 == 1
            for c in (first, first + 1):
                claimed[c] += 1
                assert (nodes[c, 0:3] >= lo).all() and (nodes[c, 4:7] <= hi).all()
                stack.append((c, depth + 1))
    assert (seen == 1).all()
    assert (claimed[1:] == 1).all()
    assert max_depth <= 64
    return max_depth


PLOC = pytest.mark.parametrize("ploc", [False, True], ids=["lbvh", "ploc"])


@PLOC
@pytest.mark.parametrize("max_leaf", [1, 4, 16])
def test_tree_is_valid_and_a_permutation(pt, max_leaf, ploc):
    sc = pt.Scene.soup(20000, seed=11, size=0.3)
    src = sc.arrays()["tris"]
    c = pt.PathTracer(8, 8)
    try:
        nodes, sorted_tris, ms = c.build_bvh_device(src, max_leaf, ploc=ploc)
    finally:
        c.close()
    assert ms > 0.0
    # the output triangles are the input triangles, re-ordered
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(key(src), key(sorted_tris))
    _check_tree(nodes, sorted_tris, max_leaf)


@PLOC
def test_duplicates_and_flat_scenes(pt, ploc):
    """Coincident centroids (identical Morton codes) and a scene with zero extent on one axis."""
    sc = pt.Scene.soup(64, seed=3, size=0.3)
    t = sc.arrays()["tris"]
    dup = np.repeat(t[:8], 100, axis=0)                # 800 triangles, 8 distinct
    flat = t.copy().view(np.float32)
    flat[:, [2, 6, 10]] = -6.0                            # all vertices in the plane z = -6
    c = pt.PathTracer(8, 8)
    try:
        for src in (dup, flat.view(np.uint32)):
            nodes, st, _ = c.build_bvh_device(src, 2, ploc=ploc)
            depth = _check_tree(nodes, st, 800)           # duplicates cannot be split by position: leaves may not shrink...
            assert depth <= 64
    finally:
        c.close()


def test_argument_checks(pt):
    c = pt.PathTracer(8, 8)
    try:
        t = pt.Scene.soup(16, seed=1).arrays()["tris"]
        with pytest.raises(pt.PtError):
            c.build_bvh_device(t[:1], 4)
        with pytest.raises(pt.PtError):
            c.build_bvh_device(t, 17)
    finally:
        c.close()


@PLOC
@pytest.mark.parametrize("maker", ["soup", "bunny"])
def test_closest_hits_through_the_gpu_tree(pt, ol, maker, ploc):
    sah = pt.Scene.soup(20000, seed=11, size=0.3) if maker == "soup" else pt.Scene.bunny(spheres=True)
    lbvh = pt.Scene.soup(20000, seed=11, size=0.3) if maker == "soup" else pt.Scene.bunny(spheres=True)
    ro, rd = _rays(200000, 5)
    c = pt.PathTracer(8, 8)
    try:
        c.finalize_scene_device(lbvh, 4, ploc=ploc)
        assert lbvh.hash() != sah.hash()
        c.upload_scene(lbvh)
        prim, th, kind, uv = c.trace_closest(ro, rd)
        c.upload_scene(sah)
        sprim, sth, skind, suv = c.trace_closest(ro, rd)
    finally:
        c.close()
    # bit-exact against the oracle walking the same (GPU-built) tree
    osc = ol.scene_from_desc(lbvh.desc())
    oprim, oth, okind, ouv = ol.trace_closest(osc, ro, rd)
    assert np.array_equal(kind, okind) and np.array_equal(prim, oprim) and np.array_equal(th, oth)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32))
    # and the same surface as through the Sweep-SAH tree: same hit/miss, same fp16 distance except where the
    # order-dependent quantisation of tmax (fc:375) lets the visiting order pick between near-coincident hits
    assert np.array_equal(kind != 0, skind != 0)
    same_t = (th == sth).mean()
    assert same_t > 0.995, same_t
    ulp = np.abs(th.astype(np.int32) - sth.astype(np.int32))
    assert ulp.max() <= 2, ulp.max()
    # identical triangle wherever the distance agrees, up to exact ties: compare the triangles' bytes, not their indices
    a = lbvh.arrays()["tris"]
    b = sah.arrays()["tris"]
    tri_hit = (kind == 1) & (skind == 1) & (th == sth)
    same_tri = (a[prim[tri_hit]] == b[sprim[tri_hit]]).all(axis=1).mean()
    assert same_tri > 0.995, same_tri


@PLOC
def test_frames_through_the_gpu_tree_match_the_oracle(pt, ol, ploc):
    sc = pt.Scene.bunny(freq=12, target_tris=0, seed=2)
    c = pt.PathTracer(64, 64)
    try:
        c.finalize_scene_device(sc, 4, ploc=ploc)
        c.upload_scene(sc)
        c.set_camera(pt.default_camera())
        seeds = pt.frame_seeds(0, 3)
        c.render(0, seeds, bounces=5)
        img = c.read_framebuffer()
    finally:
        c.close()
    osc = ol.scene_from_desc(sc.desc())
    ref, _ = ol.render(osc, pt.default_camera(), 64, 64, seeds, bounces=5)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


@PLOC
def test_million_triangles_build_time(pt, ploc):
    """C5-class set-up: 1 M triangles built on the device, uploaded and traced; prints the build time next to the CPU builder's."""
    import time
    rng = np.random.default_rng(9)
    n = 1000000
    base = pt.Scene.soup(4, seed=1).arrays()["tris"]
    tris = np.repeat(base[:1], n, axis=0).view(np.float32).copy()
    cen = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    off = rng.uniform(-0.01, 0.01, (n, 3, 3)).astype(np.float32)
    for k in range(3):
        tris[:, 4 * k:4 * k + 3] = cen + off[:, k]
    c = pt.PathTracer(8, 8)
    try:
        t0 = time.time()
        nodes, st, ms = c.build_bvh_device(tris.view(np.uint32), 4, ploc=ploc)
        wall = time.time() - t0
    finally:
        c.close()
    depth = _check_tree(nodes, st, 4)
    print(f"\n1M triangles: {'PLOC' if ploc else 'LBVH'} build {ms:.2f} ms on the device ({wall * 1e3:.0f} ms incl. PCIe both ways), {nodes.shape[0]} nodes, depth {depth}")
