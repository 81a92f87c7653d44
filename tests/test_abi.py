"""The C-ABI library loads and exports every symbol include/*.h declares; argument checking that needs no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void|uint32_t|uint64_t|char|pth_scene|pth_model|pt_ctx)\s*\*?\s*(pth?_[a-z0-9_]+)\s*\(", re.M)


def _declared():
    names = []
    for h in ("pt355.h", "pt355_host.h"):
        names += DECL.findall(open(os.path.join(ROOT, "include", h)).read())
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound(pt):
    L = pt.lib()
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/ but not exported by libpt355.so"
        assert n in pt.EXPORTS, f"{n} is declared in include/ but has no ctypes prototype"
    for n in pt.EXPORTS:
        assert n in names, f"{n} is bound in Python but not declared in include/"


def test_struct_layouts_match_the_header(pt):
    assert ctypes.sizeof(pt.pt_config) == 64
    assert ctypes.sizeof(pt.pt_scene_desc) == 6 * 16 + 16 + 24  # six arrays, env map, texture array + its sampler state
    assert ctypes.sizeof(pt.pt_counters) == 16 * 8


def test_library_does_not_link_the_oracle(pt):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", pt.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in out
    ldd = subprocess.run(["ldd", pt.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd and "libptref" not in ldd


def test_create_argument_checks(pt):
    L = pt.lib()
    h = ctypes.c_void_p()
    for w, hgt in ((0, 8), (12, 8), (8, 7)):
        cfg = pt.pt_config(width=w, height=hgt, device=0)
        assert L.pt_create(ctypes.byref(cfg), ctypes.byref(h)) == 1  # PT_EINVAL
        assert b"multiples of 8" in L.pt_last_error(None)
    cfg = pt.pt_config(width=8, height=8, device=0, rank=2, world=2)
    assert L.pt_create(ctypes.byref(cfg), ctypes.byref(h)) == 1
    cfg = pt.pt_config(width=8, height=8, device=0, partition=7)
    assert L.pt_create(ctypes.byref(cfg), ctypes.byref(h)) == 1
    assert L.pt_create(None, ctypes.byref(h)) == 1


def test_no_cpu_fallback(pt):
    """Without a HIP device the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError, match="no CPU fallback"):
        pt.PathTracer(64, 64)


def test_tonemap_matches_the_blit_shader(pt):
    # shaders/framebuffer.frag:12-25: NaN -> red, Inf -> green, else pow(rgb, 1/2.2); row 0 of the image is the bottom
    img = np.zeros((2, 3, 4), np.float32)
    img[0, 0, :3] = [np.nan, 0.5, 0.5]
    img[0, 1, :3] = [0.5, np.inf, 0.5]
    img[0, 2, :3] = [0.0, 0.25, 1.0]
    img[1, 0, :3] = [4.0, 1.0, 0.5]
    out = pt.tonemap_rgb8(img)
    assert out.shape == (2, 3, 3)
    assert out[1, 0].tolist() == [255, 0, 0] and out[1, 1].tolist() == [0, 255, 0]
    want = np.clip(np.round(np.power(np.array([0.0, 0.25, 1.0]), 1 / 2.2) * 255), 0, 255)
    assert np.abs(out[1, 2].astype(int) - want).max() <= 1
    assert out[0, 0].tolist()[0] == 255 and out[0, 0].tolist()[1] == 255


def test_image_files_round_trip(pt, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 4, (24, 40, 4)).astype(np.float32)
    img[..., 3] = 1.0
    pfm = str(tmp_path / "a.pfm")
    pt.write_pfm(pfm, img)
    back = pt.read_pfm(pfm, 40, 24)
    assert np.array_equal(back.view(np.uint32), img.view(np.uint32))
    with pytest.raises(pt.PtError):
        pt.read_pfm(pfm, 41, 24)
    ppm = str(tmp_path / "a.ppm")
    pt.write_ppm(ppm, img)
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n40 24\n255\n") and len(raw) == len(b"P6\n40 24\n255\n") + 40 * 24 * 3
    assert np.array_equal(np.frombuffer(raw[-40 * 24 * 3:], np.uint8).reshape(24, 40, 3), pt.tonemap_rgb8(img))
    # the diff tool reads what the library writes
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "imgdiff.py"), pfm, pfm], capture_output=True, text=True)
    assert out.returncode == 0 and "bit-identical 100.0000 %" in out.stdout
