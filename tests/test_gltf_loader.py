"""Row L: the glTF loader + Model (pth_load_gltf, pth_model_*) against the REFERENCE'S OWN LoadGLTF / Model
(src/loader.cpp, src/scene/model.cpp compiled from /root/reference into oracle/_ref; outputs committed as
tests/golden/ref_gltf.npz by tests/golden/make_golden_gltf.py) on the texture-less .glb files under tests/golden/gltf/.
Bit-exact: fat triangles, material indices, MaterialGLSL bytes, TriangleGLSL bytes, as loaded and after the scene set-up
of src/main.cpp:26-29.  The textured path cannot run through the reference's loader without a GL context; its two image
steps (decode of PNG / JPEG, resize to 512 x 512) equal the reference's own stb build byte for byte: against committed
outputs of that build (tests/golden/ref_jpeg.npz, ref_resize.npz) everywhere, and live where the build exists."""
import ctypes
import os
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLTF = os.path.join(ROOT, "tests", "golden", "gltf")
CASES = ["single", "multiprim", "nodes", "interleaved", "cornell_lit"]


@pytest.fixture(scope="module")
def G(golden):
    return golden("ref_gltf.npz")


def _load(pt, name, ops=None):
    m = pt.Model(os.path.join(GLTF, name + ".glb"))
    for op in (ops if ops is not None else []):
        kind, v = int(op[0]), op[1:4]
        {0: m.translate, 1: m.rotate, 2: m.scale, 3: lambda _: m.apply_matrix()}[kind](v)
    return m


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag", ["loaded", "main"])
def test_loader_output_equals_the_reference_loader(pt, G, name, tag):
    m = _load(pt, name, G["main_ops"] if tag == "main" else None)
    a = m.arrays()
    fat = G[f"{name}_{tag}_fat"]
    assert a["v"].shape[0] == fat.shape[0]
    assert np.array_equal(a["v"].view(np.uint32), fat[:, 0:9].view(np.uint32))
    assert np.array_equal(a["n"].view(np.uint32), fat[:, 9:18].view(np.uint32))
    assert np.array_equal(a["uv"].view(np.uint32), fat[:, 18:24].view(np.uint32))
    assert np.array_equal(a["mat"], G[f"{name}_{tag}_mat"])
    assert np.array_equal(a["materials"], G[f"{name}_{tag}_mats"])
    # Model::ConvertToSSBOFormat (src/scene/model.cpp:15-25): the packed 64-byte records
    packed = np.zeros((fat.shape[0], 16), np.uint32)
    for i in range(fat.shape[0]):
        pt.lib().pth_pack_triangle(a["v"][i].ctypes.data, a["uv"][i].ctypes.data, a["n"][i].ctypes.data, int(a["mat"][i]), packed[i].ctypes.data)
    assert np.array_equal(packed, G[f"{name}_{tag}_tris"])


def test_quirks_are_the_reference_s(pt, G):
    # multi-primitive mesh: primitive k re-emits the triangles of primitives 0..k-1
    a = _load(pt, "multiprim").arrays()
    assert a["v"].shape[0] == 12 + 24 + 36
    assert np.array_equal(a["v"][0:12], a["v"][12:24]) and not np.array_equal(a["mat"][0:12], a["mat"][12:24])
    mats = a["materials"].view(np.float32)
    # item 13: the metallic material comes out Lambertian (type 0) with roughness 2.0; the emissive one is a LIGHT
    assert mats[1, 7] == 0.0 and mats[1, 3] == 2.0
    assert mats[2, 7] == -1.0 and np.allclose(mats[2, 8:11], [12.0, 6.0, 3.0])
    # default material (no material on the primitive): magenta Lambert
    mats = _load(pt, "nodes").arrays()["materials"].view(np.float32)
    assert mats[1, 0:3].tolist() == [1.0, 0.0, 1.0]
    # rough dielectric -> Oren-Nayar with roughness * 0.35
    mats = _load(pt, "single").arrays()["materials"].view(np.float32)
    assert mats[0, 7] == 1.0 and mats[0, 3] == np.float32(0.5) * np.float32(0.35)


def test_scene_from_gltf_is_what_main_builds(pt, G, ol):
    # src/main.cpp:18-63 around the Cornell-shaped file: model set-up, four gold spheres, CalculateBVH, emissive lists
    sc = pt.Scene.from_gltf(os.path.join(GLTF, "cornell_lit.glb"))
    a = sc.arrays()
    ref_tris = G["cornell_lit_main_tris"]
    assert a["tris"].shape[0] == ref_tris.shape[0] == 66
    # the BVH reorders; as multisets the packed triangles are the reference's
    assert sorted(map(bytes, a["tris"])) == sorted(map(bytes, ref_tris))
    assert a["materials"].shape[0] == 7 + 4 and np.array_equal(a["materials"][:7], G["cornell_lit_main_mats"])
    assert a["spheres"].shape[0] == 4 and a["light_tris"].size > 0
    # and the oracle can render it (plumbing check of the six arrays)
    img, cnt = ol.render(ol.scene_from_desc(sc.desc()), pt.default_camera(), 32, 32, pt.frame_seeds(0, 2))
    assert np.isfinite(img).all() and img[..., :3].max() > 0 and cnt["rays_shadow"] > 0


def test_errors_where_the_reference_errors_or_crashes(pt, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gltf_writer as gw
    with pytest.raises(pt.PtError, match="cannot read"):
        pt.Model(str(tmp_path / "missing.glb"))
    p, n, uv, i = gw.box((0, 0, 0), (1, 1, 1))
    # transparent base colour (loader.cpp:192-197)
    g = gw.Glb()
    g.node(mesh=g.mesh([g.primitive(p, n, i, uv, g.material(base=(1, 1, 1, 0.5), metallic=0.0, roughness=0.0))]))
    f = tmp_path / "alpha.glb"
    f.write_bytes(g.tobytes())
    with pytest.raises(pt.PtError, match="transparency"):
        pt.Model(str(f))
    # 32-bit indices (loader.cpp:146-152)
    g = gw.Glb()
    prim = g.primitive(p, n, i, uv, None)
    iv = g.view(i.astype(np.uint32).tobytes(), target=34963)
    prim["indices"] = g.accessor(iv, 5125, i.size, "SCALAR")
    g.node(mesh=g.mesh([prim]))
    f = tmp_path / "u32.glb"
    f.write_bytes(g.tobytes())
    with pytest.raises(pt.PtError, match="uint16"):
        pt.Model(str(f))
    # points instead of triangles (loader.cpp:73-79)
    g = gw.Glb()
    prim = g.primitive(p, n, i, uv, None)
    prim["mode"] = 0
    g.node(mesh=g.mesh([prim]))
    f = tmp_path / "points.glb"
    f.write_bytes(g.tobytes())
    with pytest.raises(pt.PtError, match="triangles"):
        pt.Model(str(f))
    # no normals: the reference reads an empty array; here an error
    g = gw.Glb()
    prim = g.primitive(p, n, i, uv, None)
    del prim["attributes"]["NORMAL"]
    g.node(mesh=g.mesh([prim]))
    f = tmp_path / "nonormals.glb"
    f.write_bytes(g.tobytes())
    with pytest.raises(pt.PtError, match="normals"):
        pt.Model(str(f))
    # garbage
    f = tmp_path / "garbage.glb"
    f.write_bytes(b"glTF\x02\x00\x00\x00" + b"\x00" * 40)
    with pytest.raises(pt.PtError):
        pt.Model(str(f))


def test_gltf_json_with_data_uri_and_external_bin(pt, tmp_path):
    # .gltf (JSON) with the buffer as a base64 data URI, and as an external .bin: same triangles as the .glb
    import base64
    import json
    blob = open(os.path.join(GLTF, "single.glb"), "rb").read()
    jlen = struct.unpack("<I", blob[12:16])[0]
    j = json.loads(blob[20:20 + jlen])
    binary = blob[20 + jlen + 8:]
    want = _load(pt, "single").arrays()
    j["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(binary).decode()
    (tmp_path / "a.gltf").write_text(json.dumps(j, indent=2))
    got = pt.Model(str(tmp_path / "a.gltf")).arrays()
    assert all(np.array_equal(got[k], want[k]) for k in want)
    j["buffers"][0]["uri"] = "a.bin"
    (tmp_path / "a.bin").write_bytes(binary)
    (tmp_path / "b.gltf").write_text(json.dumps(j))
    got = pt.Model(str(tmp_path / "b.gltf")).arrays()
    assert all(np.array_equal(got[k], want[k]) for k in want)


# ------------------------------------------------------------------ textured path
def _png(rgb, color_type=2, filt=None):
    """Own minimal PNG writer (8-bit; colour type 2 RGB, 6 RGBA, 0 grey); filt: per-row filter types to exercise the decoder."""
    h, w = rgb.shape[:2]
    raw = bytearray()
    arr = rgb.astype(np.uint8)
    bpp = arr.shape[2] if arr.ndim == 3 else 1
    rows = arr.reshape(h, w * bpp).astype(np.int32)
    for y in range(h):
        f = 0 if filt is None else filt[y % len(filt)]
        cur = rows[y]
        up = rows[y - 1] if y else np.zeros_like(cur)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
        if f == 0:
            out = cur
        elif f == 1:
            out = cur - left
        elif f == 2:
            out = cur - up
        elif f == 3:
            out = cur - ((left + up) >> 1)
        else:
            p = left + up - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
            out = cur - pred
        raw.append(f)
        raw += (out & 0xff).astype(np.uint8).tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def _decode(pt, png):
    buf = np.frombuffer(png, np.uint8)
    w, h, ch = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert pt.lib().pth_decode_image_rgb8(buf.ctypes.data, buf.size, None, 0, ctypes.byref(w), ctypes.byref(h), ctypes.byref(ch)) == 0
    out = np.zeros((h.value, w.value, 3), np.uint8)
    assert pt.lib().pth_decode_image_rgb8(buf.ctypes.data, buf.size, out.ctypes.data, out.size, None, None, None) == 0
    return out, ch.value


def test_png_decoder(pt, ol):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    got, ch = _decode(pt, _png(img, 2, filt=[0, 1, 2, 3, 4]))
    assert ch == 3 and np.array_equal(got, img)
    rgba = rng.integers(0, 256, (16, 20, 4), dtype=np.uint8)
    got, ch = _decode(pt, _png(rgba, 6, filt=[4, 3]))
    assert ch == 4 and np.array_equal(got, rgba[..., :3])          # alpha dropped, like stbi with req_comp = 3
    grey = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    got, ch = _decode(pt, _png(grey, 0, filt=[1]))
    assert ch == 1 and np.array_equal(got, np.repeat(grey[..., None], 3, 2))
    # against the reference's own stb_image where it exists (build container)
    R = ol.load_ref()
    if R is not None and hasattr(R, "ref_stbi_load_rgb8"):
        for im, ct in ((img, 2), (rgba, 6), (grey, 0)):
            png = np.frombuffer(_png(im, ct, filt=[0, 1, 2, 3, 4]), np.uint8)
            h, w = im.shape[:2]
            out = np.zeros((h, w, 3), np.uint8)
            ww, hh, cc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            assert R.ref_stbi_load_rgb8(ctypes.c_void_p(png.ctypes.data), ctypes.c_size_t(png.size), ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(out.size), ctypes.byref(ww), ctypes.byref(hh), ctypes.byref(cc)) == 0
            mine, ch = _decode(pt, png.tobytes())
            assert (ww.value, hh.value, cc.value) == (w, h, ch) and np.array_equal(mine, out)


JPEG = os.path.join(ROOT, "tests", "golden", "jpeg")
JPEG_CASES = sorted(f[:-4] for f in os.listdir(JPEG) if f.endswith(".jpg"))


@pytest.mark.parametrize("name", JPEG_CASES)
def test_jpeg_decoder_equals_the_reference_s_stb(pt, ol, golden, name):
    """tests/golden/jpeg/*.jpg (baseline / progressive / 4:4:4, 4:2:2, 4:2:0, 4:1:1, 4:4:0, mixed and 4x2 sampling /
    greyscale / restart intervals / optimised tables / 'RGB' ids / Adobe markers / 16-bit tables whose products wrap /
    1-pixel-wide edge cases): every byte equals what the reference's own stb_image build returned for the file
    (tests/golden/ref_jpeg.npz, made by tests/golden/make_golden_jpeg.py), and again live where that build exists."""
    G = golden("ref_jpeg.npz")
    data = open(os.path.join(JPEG, name + ".jpg"), "rb").read()
    got, ch = _decode(pt, data)
    assert ch == int(G[name + "_channels"]) and got.shape == G[name].shape and np.array_equal(got, G[name])
    R = ol.load_ref()
    if R is not None and hasattr(R, "ref_stbi_load_rgb8"):
        buf = np.frombuffer(data, np.uint8)
        out = np.zeros_like(got)
        ww, hh, cc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert R.ref_stbi_load_rgb8(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.size), ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(out.size), ctypes.byref(ww), ctypes.byref(hh), ctypes.byref(cc)) == 0
        assert np.array_equal(out, got) and cc.value == ch


def test_jpeg_decoder_on_fresh_files(pt, ol):
    """files written on the spot by tests/jpeg_writer.py: the decoded picture is the encoded one to within the
    quantisation (so the fixtures are pictures, not noise that two decoders happen to agree on), equals stb's where the
    reference build exists, and damaged / unsupported files are refused with a message"""
    import jpeg_writer as jw
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:40, 0:52]
    img = np.stack([xx * 4, yy * 6, 255 - xx * 2 - yy * 2], 2).clip(0, 255).astype(np.uint8)
    R = ol.load_ref()
    for sampling in ([(1, 1)] * 3, [(2, 2), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(2, 2), (2, 1), (1, 2)]):
        hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
        planes = [jw.subsample(c, hmax // s[0], vmax // s[1]) for c, s in zip(jw.rgb_to_ycc(img), sampling)]
        data = jw.encode(planes, sampling, 52, 40, quant=[2] * 64, restart=int(rng.integers(0, 4)))
        got, ch = _decode(pt, data)
        assert ch == 3 and np.abs(got.astype(int) - img.astype(int)).mean() < 3.0, sampling
        if R is not None and hasattr(R, "ref_stbi_load_rgb8"):
            buf = np.frombuffer(data, np.uint8)
            out = np.zeros_like(got)
            ww, hh, cc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            assert R.ref_stbi_load_rgb8(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.size), ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(out.size), ctypes.byref(ww), ctypes.byref(hh), ctypes.byref(cc)) == 0
            assert np.array_equal(out, got)
    good = jw.encode([img[..., 1]], [(1, 1)], 52, 40)
    got, ch = _decode(pt, good)
    assert ch == 1 and (got[..., 0] == got[..., 2]).all() and np.abs(got[..., 0].astype(int) - img[..., 1]).mean() < 4.0
    w = ctypes.c_int()
    cmyk = jw.encode([img[..., 0], img[..., 1], img[..., 2], img[..., 0]], [(1, 1)] * 4, 52, 40, ids=(1, 2, 3, 4))
    for bad, what in ((cmyk, "CMYK"), (good[:200].replace(b"\xff\xc0", b"\xff\xc9"), "arithmetic"), (b"\xff\xd8\xff\xd9", "SOF"), (b"GIF89a" + bytes(20), "neither")):
        buf = np.frombuffer(bad, np.uint8)
        assert pt.lib().pth_decode_image_rgb8(buf.ctypes.data, buf.size, None, 0, ctypes.byref(w), ctypes.byref(w), ctypes.byref(w)) != 0
        assert what in pt.lib().pth_last_error().decode(), (what, pt.lib().pth_last_error())


def _resize(pt, img):
    out = np.zeros((512, 512, 3), np.uint8)
    a = np.ascontiguousarray(img, np.uint8)
    assert pt.lib().pth_resize_rgb8_512(a.ctypes.data, a.shape[1], a.shape[0], out.ctypes.data) == 0
    return out


def test_resize_to_512(pt, ol, golden):
    rng = np.random.default_rng(4)
    same = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    assert np.array_equal(_resize(pt, same), same)                  # already 512 x 512: untouched (loader.cpp:222)
    flat = np.full((100, 300, 3), 77, np.uint8)
    assert (_resize(pt, flat) == 77).all()                           # weights sum to 1
    yy, xx = np.mgrid[0:256, 0:256]
    ramp = np.stack([xx, yy, (xx + yy) // 2], 2).astype(np.uint8)
    up = _resize(pt, ramp).astype(np.float64)
    want = np.stack([np.clip((np.arange(512) + 0.5) / 2 - 0.5, 0, 255)] * 512, 0)
    assert np.abs(up[..., 0] - want)[:, 4:-4].max() <= 1.0           # a linear ramp is reproduced by Catmull-Rom
    # byte for byte what the reference's own stb_image_resize build makes of the seeded images of
    # tests/golden/make_golden_resize.py (enlarging, shrinking, mixed, one axis already 512, 1 x 1 ...): CRC of the whole
    # output + every 16th pixel from tests/golden/ref_resize.npz, and the full image live where that build exists
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_resize as mg
    G = golden("ref_resize.npz")
    R = ol.load_ref()
    for i, shape in enumerate(mg.SHAPES):
        src = mg.resize_case(i)
        got = _resize(pt, src)
        assert np.array_equal(got[::16, ::16], G[f"sub_{i}"]), shape
        assert (zlib.crc32(got.tobytes()) & 0xffffffff) == int(G[f"crc_{i}"]), shape
        if R is not None and hasattr(R, "ref_stbir_resize_512"):
            ref = np.zeros((512, 512, 3), np.uint8)
            assert R.ref_stbir_resize_512(ctypes.c_void_p(src.ctypes.data), src.shape[1], src.shape[0], ctypes.c_void_p(ref.ctypes.data)) == 1
            assert np.array_equal(got, ref), shape


def test_textured_glb_fills_the_layer_array(pt, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gltf_writer as gw
    rng = np.random.default_rng(8)
    tex_a = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    tex_b = np.full((64, 32, 3), (10, 200, 90), np.uint8)
    g = gw.Glb()
    images, textures = [], []
    for im in (tex_a, tex_b):
        v = g.view(_png(im, 2, filt=[4]))
        images.append({"bufferView": v, "mimeType": "image/png"})
    g.j["images"] = images
    g.j["samplers"] = [{"magFilter": 9728, "minFilter": 9987, "wrapS": 33071, "wrapT": 33648}]
    g.j["textures"] = [{"source": 0}, {"source": 1, "sampler": 0}, {"source": 0}]   # three textures -> three layers, the third never loaded
    prims = []
    for k, ti in enumerate((0, 1)):
        p, n, uv, i = gw.box((2.0 * k, 0, 0), (0.5, 0.5, 0.5))
        m = g.material(base=(1, 1, 1, 1), metallic=0.0, roughness=0.0)
        g.j["materials"][m]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": ti}
        prims.append(g.primitive(p, n, i, uv, m))
    g.node(mesh=g.mesh(prims))
    f = tmp_path / "tex.glb"
    f.write_bytes(g.tobytes())
    m = pt.Model(str(f))
    tex, smp = m.textures()
    assert tex.shape == (3, 512, 512, 3)
    assert np.array_equal(tex[0], tex_a)                                   # 512 x 512: stored as decoded
    assert (tex[1] == np.array([10, 200, 90], np.uint8)).all()             # resized constant image
    assert (tex[2] == 0).all()                                             # allocated, never filled (glTextureStorage3D)
    assert smp == (33071, 33648, 9728)                                     # the last textured primitive's sampler wins, array-wide
    mats = m.arrays()["materials"].view(np.float32)
    assert mats[0, 11] == 0.0 and mats[1, 11] == 1.0                       # diffuse_tex_index = load order
    sc = pt.Scene.from_gltf(str(f), spheres=False)
    assert sc.textures.shape[0] == 3 and sc.tex_sampler == (33071, 33648, 9728)


def test_textured_glb_with_a_jpeg_image(pt, tmp_path):
    """image/jpeg buffer views go through the same path (stbi_load_from_memory sniffs the format, loader.cpp:209)"""
    import gltf_writer as gw
    import jpeg_writer as jw
    yy, xx = np.mgrid[0:512, 0:512]
    img = np.stack([xx // 2, yy // 2, (xx + yy) // 4], 2).astype(np.uint8)
    sampling = [(2, 2), (1, 1), (1, 1)]
    planes = [jw.subsample(c, 2 // s[0], 2 // s[1]) for c, s in zip(jw.rgb_to_ycc(img), sampling)]
    data = jw.encode(planes, sampling, 512, 512, quant=[4] * 64)
    g = gw.Glb()
    g.j["images"] = [{"bufferView": g.view(data), "mimeType": "image/jpeg"}]
    g.j["textures"] = [{"source": 0}]
    p, n, uv, i = gw.box((0, 0, 0), (0.5, 0.5, 0.5))
    m = g.material(base=(1, 1, 1, 1), metallic=0.0, roughness=0.0)
    g.j["materials"][m]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    g.node(mesh=g.mesh([g.primitive(p, n, i, uv, m)]))
    f = tmp_path / "jpg.glb"
    f.write_bytes(g.tobytes())
    tex, smp = pt.Model(str(f)).textures()
    dec, ch = _decode(pt, data)
    assert ch == 3 and tex.shape == (1, 512, 512, 3) and np.array_equal(tex[0], dec)
    assert np.abs(dec.astype(int) - img.astype(int)).mean() < 2.0
    assert smp == (10497, 10497, 9729)


# ------------------------------------------------------------------ environment map (.hdr)
def _write_hdr(path, rgbe, rle):
    """Own minimal Radiance writer: rgbe u8 [h, w, 4]; rle = new-style run-length scanlines (w in 8..32767)."""
    h, w = rgbe.shape[:2]
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            ch = rgbe[y, :, k]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and ch[i + run] == ch[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(ch[i])])
                    i += run
                else:
                    j = i
                    while j < w and j - i < 128 and not (j + 2 < w and ch[j] == ch[j + 1] == ch[j + 2]):
                        j += 1
                    j = max(j, i + 1)
                    out += bytes([j - i]) + ch[i:j].tobytes()
                    i = j
    with open(path, "wb") as f:
        f.write(bytes(out))


@pytest.mark.parametrize("rle", [False, True])
def test_env_hdr_loader(pt, ol, tmp_path, rle):
    rng = np.random.default_rng(12)
    h, w = 20, 48
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = rng.integers(0, 256, (h, w, 3))
    rgbe[..., 3] = rng.integers(118, 134, (h, w))       # exponents around 128: values from ~0.001 to ~32
    rgbe[2, 5:30] = [200, 100, 50, 129]                 # a run, so the RLE writer emits run packets
    rgbe[3, :, 3] = 0                                   # exponent 0 = black
    path = str(tmp_path / "env.hdr")
    _write_hdr(path, rgbe, rle)
    img = pt.load_env_hdr(path)
    assert img.shape == (h, w, 3)
    # closed form: v = m * 2^(e - 136); 8-bit = clamp(v * 255 + 0.5)
    v = rgbe[..., :3].astype(np.float32) * np.exp2(rgbe[..., 3:4].astype(np.float32) - 136.0).astype(np.float32)
    v[rgbe[..., 3] == 0] = 0
    want = np.clip(v * np.float32(255) + np.float32(0.5), 0, 255).astype(np.uint8)
    assert np.array_equal(img, want)
    R = ol.load_ref()
    if R is not None and hasattr(R, "ref_env_hdr_load"):   # the reference's own stb call sequence (build container)
        ref = np.zeros((h, w, 3), np.uint8)
        ww, hh = ctypes.c_int(), ctypes.c_int()
        assert R.ref_env_hdr_load(path.encode(), ctypes.c_void_p(ref.ctypes.data), ctypes.c_size_t(ref.size), ctypes.byref(ww), ctypes.byref(hh)) == 0
        assert (ww.value, hh.value) == (w, h) and np.array_equal(img, ref)
    with pytest.raises(pt.PtError, match="not a Radiance"):
        bad = tmp_path / "bad.hdr"
        bad.write_bytes(b"P6\n1 1\n255\n\0\0\0")
        pt.load_env_hdr(str(bad))
