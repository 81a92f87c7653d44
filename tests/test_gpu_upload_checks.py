"""pt_upload_scene must turn a malformed scene into an error code -- never into a GPU fault or an endless traversal
(INTEGRATION.md "Error behaviour").  Also: a foreign BVH whose sibling leaves are not contiguous in the triangle
array is traversed like the shader traverses it (left leaf's triangles, then the right leaf's)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _desc_from(pt, arrays, **override):
    a = dict(arrays)
    a.update(override)
    d = pt.pt_scene_desc()
    keep = []
    for field, count, key in (("spheres", "n_spheres", "spheres"), ("tris", "n_tris", "tris"), ("light_tris", "n_light_tris", "light_tris"),
                              ("materials", "n_materials", "materials"), ("nodes", "n_nodes", "nodes"),
                              ("light_spheres", "n_light_spheres", "light_spheres")):
        arr = np.ascontiguousarray(a[key], np.uint32)
        keep.append(arr)
        setattr(d, field, arr.ctypes.data if arr.size else None)
        setattr(d, count, arr.shape[0])
    d._keep = keep
    return d


def _upload_fails(pt, d, needle):
    c = pt.PathTracer(8, 8)
    try:
        with pytest.raises(pt.PtError) as e:
            c.upload_scene(d)
        assert needle in str(e.value), str(e.value)
    finally:
        c.close()


def test_malformed_scenes_are_error_codes(pt):
    base = pt.Scene.soup(200, seed=4).arrays()
    nodes = base["nodes"].copy()
    nf = nodes.view(np.float32)
    inner = [i for i in range(1, nodes.shape[0]) if nf[i, 7] == 0.0]
    leaf = [i for i in range(1, nodes.shape[0]) if nf[i, 7] > 0.0]

    bad = nodes.copy(); bad.view(np.float32)[inner[0], 3] = float(nodes.shape[0] + 10)          # child outside the array
    _upload_fails(pt, _desc_from(pt, base, nodes=bad), "outside the array")

    bad = nodes.copy(); bad.view(np.float32)[leaf[0], 3] = float(base["tris"].shape[0] - 1); bad.view(np.float32)[leaf[0], 7] = 5.0
    _upload_fails(pt, _desc_from(pt, base, nodes=bad), "runs past the triangle array")

    bad = nodes.copy(); bad.view(np.float32)[inner[1], 3] = nf[inner[0], 3]                       # two parents for one pair: a DAG / cycle
    _upload_fails(pt, _desc_from(pt, base, nodes=bad), "not a tree")

    bad = nodes.copy(); bad.view(np.float32)[inner[0], 3] = 1.0                                   # points back at the root pair: a cycle
    _upload_fails(pt, _desc_from(pt, base, nodes=bad), "not a tree")

    bad = nodes.copy(); bad.view(np.float32)[inner[0], 3] = np.nan
    _upload_fails(pt, _desc_from(pt, base, nodes=bad), "negative/NaN")

    tris = base["tris"].copy(); tris[3, 15] = 99
    _upload_fails(pt, _desc_from(pt, base, tris=tris), "uses material")

    _upload_fails(pt, _desc_from(pt, base, nodes=nodes[:-1]), "odd node count")
    _upload_fails(pt, _desc_from(pt, base, light_tris=np.array([10 ** 6], np.uint32)), "light triangle index")


def test_a_chain_deeper_than_the_stack_is_refused(pt):
    """A degenerate 70-level chain (every right child a one-triangle leaf) is beyond the 64-entry traversal stack."""
    base = pt.Scene.soup(80, seed=4).arrays()
    tris = base["tris"]
    levels = 70
    nodes = np.zeros((2 * levels + 1, 8), np.float32)
    lo, hi = np.float32(-10.0), np.float32(10.0)
    nodes[:, 0:3] = lo; nodes[:, 4:7] = hi
    nodes[0, 3] = 1.0
    for k in range(levels):
        l, r = 1 + 2 * k, 2 + 2 * k
        if k + 1 < levels:
            nodes[l, 3] = 1 + 2 * (k + 1); nodes[l, 7] = 0.0
        else:
            nodes[l, 3] = 0.0; nodes[l, 7] = 1.0
        nodes[r, 3] = 1 + (k % 70); nodes[r, 7] = 1.0
    _upload_fails(pt, _desc_from(pt, base, nodes=nodes.view(np.uint32), tris=tris), "deeper than")


def test_the_deepest_accepted_tree_fills_the_whole_stack(pt, ol):
    """63 levels of pairs whose two children are both inner nodes with the scene's bounds (one continues the chain, the other
    holds two one-triangle leaves): every ray that enters the box pushes at every level, so each lane's stack goes through
    its LDS levels, through the whole spill tail and back down to the sentinel below its first entry.  Same hits as the oracle."""
    base = pt.Scene.soup(128, seed=21, size=0.6).arrays()
    tris = base["tris"]
    v = tris.view(np.float32)[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3)
    lo, hi = v.min(0) - np.float32(0.01), v.max(0) + np.float32(0.01)
    levels = 63
    nodes = np.zeros((1 + 2 * levels + 4 * levels + 2, 8), np.float32)
    nodes[:, 0:3] = lo; nodes[:, 4:7] = hi
    nodes[0, 3] = 1.0
    nxt = 1 + 2 * levels                      # where the leaf pairs start
    t = 0
    for k in range(levels):
        a, b = 1 + 2 * k, 2 + 2 * k          # a continues the chain (or ends it in two leaves), b holds two leaves
        for inner_node in ((b,) if k + 1 < levels else (a, b)):
            nodes[inner_node, 3], nodes[inner_node, 7] = float(nxt), 0.0
            for leaf in (nxt, nxt + 1):
                nodes[leaf, 3], nodes[leaf, 7] = float(t), 1.0
                t += 1
            nxt += 2
        if k + 1 < levels:
            nodes[a, 3], nodes[a, 7] = float(1 + 2 * (k + 1)), 0.0
    nodes = nodes[:nxt]
    assert t == 128 and nodes.shape[0] % 2 == 1
    d = _desc_from(pt, base, nodes=nodes.view(np.uint32), light_tris=np.zeros(0, np.uint32), spheres=np.zeros((0, 8), np.uint32),
                   light_spheres=np.zeros(0, np.uint32))
    rng = np.random.default_rng(3)
    n = 20000
    ro = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(d)
        c.reset_counters()
        prim, th, kind, uv = c.trace_closest(ro, rd)
        spills = c.counters()["stack_spills"]
    finally:
        c.close()
    oprim, oth, okind, ouv = ol.trace_closest(ol.scene_from_desc(d), ro, rd)
    assert (okind == 1).mean() > 0.05
    assert spills > 1000 * 40, spills      # the rays that enter the box spill some 50 entries each
    assert np.array_equal(kind, okind) and np.array_equal(prim, oprim) and np.array_equal(th, oth)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32))


def test_non_contiguous_sibling_leaves(pt, ol):
    """Foreign BVH: the two leaves of one pair point at triangle ranges that are far apart in the array."""
    base = pt.Scene.soup(64, seed=6, size=0.8).arrays()
    tris = base["tris"]
    v = tris.view(np.float32)[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3, 3)

    def box(ids):
        return v[ids].reshape(-1, 3).min(0), v[ids].reshape(-1, 3).max(0)

    groups = [list(range(0, 16)), list(range(48, 64)), list(range(16, 32)), list(range(32, 48))]  # leaf ranges: 0-15, 48-63 | 16-31, 32-47
    nodes = np.zeros((7, 8), np.float32)
    alllo, allhi = box(list(range(64)))
    nodes[0, 0:3], nodes[0, 4:7], nodes[0, 3] = alllo, allhi, 1.0
    for k, (a, b) in enumerate(((0, 1), (2, 3))):
        lo, hi = box(groups[a] + groups[b])
        nodes[1 + k, 0:3], nodes[1 + k, 4:7], nodes[1 + k, 3], nodes[1 + k, 7] = lo, hi, 3 + 2 * k, 0.0
        for j, g in enumerate((a, b)):
            glo, ghi = box(groups[g])
            i = 3 + 2 * k + j
            nodes[i, 0:3], nodes[i, 4:7], nodes[i, 3], nodes[i, 7] = glo, ghi, float(groups[g][0]), 16.0
    d = _desc_from(pt, base, nodes=nodes.view(np.uint32), light_tris=np.zeros(0, np.uint32))
    rng = np.random.default_rng(2)
    n = 50000
    ro = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(d)
        prim, th, kind, uv = c.trace_closest(ro, rd)
    finally:
        c.close()
    oprim, oth, okind, ouv = ol.trace_closest(ol.scene_from_desc(d), ro, rd)
    assert (kind == 1).mean() > 0.2
    assert np.array_equal(kind, okind) and np.array_equal(prim, oprim) and np.array_equal(th, oth)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32))


def test_astronomic_coordinates_take_the_ieee_division(pt, ol):
    """The triangle test's 1/det is v_rcp_f32 + one Newton step only when pt_upload_scene has proved |det| < 2^126 for every
    triangle (where that form equals the IEEE division bit for bit; above it the short form is wrong for EVERY input).
    Four triangles with edges of ~2^63 in front of the origin give determinants of 2^126 ... 2^127.6 and perfectly ordinary
    hits at t = 1 ... 4: the proof fails, the kernel must divide, and the hits must equal the oracle's."""
    base = pt.Scene.soup(64, seed=9).arrays()
    big = 2.0 ** 62
    tris = base["tris"][:4].copy()
    tf = tris.view(np.float32)
    for i, (s, z) in enumerate(((1.5, -1.0), (1.25, -2.0), (1.75, -3.0), (1.0, -4.0))):
        tf[i, 0:3] = (-s * big, -s * big, z)
        tf[i, 4:7] = (s * big, -s * big, z)
        tf[i, 8:11] = (0.0, s * big, z)
    nodes = np.zeros((3, 8), np.float32)
    nodes[0, 0:3], nodes[0, 4:7], nodes[0, 3] = (-2 * big, -2 * big, -4.0), (2 * big, 2 * big, -1.0), 1.0
    nodes[1, 0:3], nodes[1, 4:7], nodes[1, 3], nodes[1, 7] = (-2 * big, -2 * big, -2.0), (2 * big, 2 * big, -1.0), 0.0, 2.0
    nodes[2, 0:3], nodes[2, 4:7], nodes[2, 3], nodes[2, 7] = (-2 * big, -2 * big, -4.0), (2 * big, 2 * big, -3.0), 2.0, 2.0
    d = _desc_from(pt, base, tris=tris, nodes=nodes.view(np.uint32), light_tris=np.zeros(0, np.uint32), spheres=np.zeros((0, 8), np.uint32),
                   light_spheres=np.zeros(0, np.uint32))
    rng = np.random.default_rng(12)
    n = 20000
    ro = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-3.5, 5.0, n)], 1).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd[:, 2] = -np.abs(rd[:, 2]) - 0.2
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(d)
        prim, th, kind, uv = c.trace_closest(ro, rd)
    finally:
        c.close()
    oprim, oth, okind, ouv = ol.trace_closest(ol.scene_from_desc(d), ro, rd)
    assert (okind == 1).mean() > 0.5 and len(set(oprim[okind == 1].tolist())) == 4
    assert np.array_equal(kind, okind) and np.array_equal(prim, oprim) and np.array_equal(th, oth)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32))


def test_create_says_how_much_memory_it_needs(pt):
    """An image whose queues cannot fit (16384 x 16384 = the largest size pt_create accepts, 4 pipelines) is refused with the
    figures, before any allocation fails half way; the same image with a one-frame batch and one pipeline is accepted
    or refused by the same rule, never by a raw hipMalloc error."""
    with pytest.raises(pt.PtError, match="needs .* GB of device memory"):
        pt.PathTracer(16384, 16384, max_batch=32, pipelines=4)
    c = pt.PathTracer(2048, 2048, max_batch=1)
    c.close()
