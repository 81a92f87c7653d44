"""Host sanitizer + fuzz target: the library parses untrusted bytes itself -- GLB / JSON,
PNG, JPEG, Radiance HDR, PFM (csrc/host/gltf.cpp, jpeg.cpp) -- where the reference leans on cgltf / stb and has no sanitizer
target (CMakeLists.txt:72-81).  `make -C pathtracer_amd/csrc host-asan` builds the host translation units with
AddressSanitizer + UBSan around tools/host_fuzz.cpp; this test runs it with a fixed seed over the committed .glb / .jpg
fixtures and over PNG / HDR / PFM / textured-GLB files written on the spot.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "pathtracer_amd", "host_fuzz_asan")


@pytest.fixture(scope="module")
def fuzz_bin():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pathtracer_amd", "csrc"), "host-asan"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(FUZZ):
        pytest.skip("host-asan target did not build here (no sanitizer runtime?): " + r.stderr[-300:])
    return FUZZ


def _run(fuzz_bin, seed, per_file, files, timeout=600):
    env = dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    env.pop("LD_PRELOAD", None)  # ASan insists on being the first library in the process
    r = subprocess.run([fuzz_bin, str(seed), str(per_file)] + list(files), capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "no sanitizer report" in r.stdout
    return r.stdout


def test_fuzz_committed_glb_and_jpeg_fixtures(fuzz_bin):
    gl = sorted(os.path.join(ROOT, "tests", "golden", "gltf", f) for f in os.listdir(os.path.join(ROOT, "tests", "golden", "gltf")))
    jp = sorted(os.path.join(ROOT, "tests", "golden", "jpeg", f) for f in os.listdir(os.path.join(ROOT, "tests", "golden", "jpeg")))
    out = _run(fuzz_bin, 20261111, 500, gl)
    assert "2505 inputs" in out, out
    out = _run(fuzz_bin, 20261111, 40, jp)
    assert f"{len(jp) * 41} inputs" in out, out


def test_fuzz_generated_png_hdr_pfm_and_textured_glb(fuzz_bin, pt, tmp_path):
    import gltf_writer as gw
    from test_gltf_loader import _png, _write_hdr
    rng = np.random.default_rng(4)
    files = []
    for name, im, ct, filt in (("rgb.png", rng.integers(0, 256, (24, 40, 3)), 2, [0, 1, 2, 3, 4]), ("rgba.png", rng.integers(0, 256, (17, 9, 4)), 6, [4, 3]),
                               ("grey.png", rng.integers(0, 256, (33, 5)), 0, [1])):
        p = tmp_path / name
        p.write_bytes(_png(im, ct, filt=filt))
        files.append(str(p))
    for name, rle in (("flat.hdr", False), ("rle.hdr", True)):
        rgbe = rng.integers(0, 256, (6, 16, 4)).astype(np.uint8)
        rgbe[..., 3] = rng.integers(120, 136, (6, 16))
        _write_hdr(str(tmp_path / name), rgbe, rle)
        files.append(str(tmp_path / name))
    pt.write_pfm(str(tmp_path / "img.pfm"), rng.random((8, 8, 4), dtype=np.float32))
    files.append(str(tmp_path / "img.pfm"))
    # a GLB with an embedded PNG texture: JSON, accessors, image decode and the resize to 512 x 512 in one file
    g = gw.Glb()
    v = g.view(_png(rng.integers(0, 256, (8, 8, 3)), 2, filt=[4]))
    g.j["images"] = [{"bufferView": v, "mimeType": "image/png"}]
    g.j["samplers"] = [{"magFilter": 9728, "minFilter": 9987, "wrapS": 33071, "wrapT": 33648}]
    g.j["textures"] = [{"source": 0, "sampler": 0}]
    p, n, uv, i = gw.box((0, 0, 0), (0.5, 0.5, 0.5))
    m = g.material(base=(1, 1, 1, 1), metallic=0.0, roughness=0.0)
    g.j["materials"][m]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    g.node(mesh=g.mesh([g.primitive(p, n, i, uv, m)]))
    (tmp_path / "tex.glb").write_bytes(g.tobytes())
    files.append(str(tmp_path / "tex.glb"))
    out = _run(fuzz_bin, 7, 300, files)
    assert f"{len(files) * 301} inputs" in out, out


def test_glb_without_meshes_loads_as_an_empty_model(pt, tmp_path):
    """A glTF that holds materials but no mesh is a valid file: the model is empty, adding it to a scene adds the materials
    and no triangles (the advisor's finding: &v[0] on an empty vector)."""
    import gltf_writer as gw
    g = gw.Glb()
    g.material(base=(1, 0, 0, 1), metallic=0.0, roughness=0.5)
    f = tmp_path / "nomesh.glb"
    f.write_bytes(g.tobytes())
    m = pt.Model(str(f))
    a = m.arrays()
    assert a["v"].shape[0] == 0
    s = pt.Scene()
    s.add_model(m)                      # no triangles to add: returns without touching the empty arrays
    empty = pt.Model()                  # a freshly created model, nothing loaded
    s.add_model(empty)
    with pytest.raises(pt.PtError):
        s.finalize()                    # a scene without triangles cannot be finalized (the reference reads node 1 of none)
