"""Writes the small JPEG files under tests/golden/jpeg/ (Pillow's libjpeg encoder; any encoder would do, they are only
inputs) and tests/golden/ref_jpeg.npz = what the REFERENCE'S OWN stb_image build (oracle/_ref/libptref.so,
ref_stbi_load_rgb8 = stbi_load_from_memory(..., 3) as src/loader.cpp:209-214 calls it) decodes them to.  Run in the build
container (needs /root/reference for `make -C oracle ref`, and Pillow):  python tests/golden/make_golden_jpeg.py"""
import ctypes
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def picture(h, w, seed):
    """smooth colour gradients + an edge + a little noise: exercises DC prediction, AC runs and the chroma filters"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 128 + 100 * np.sin(xx / 5.0 + seed) * np.cos(yy / 7.0)
    g = 255 * xx / max(w - 1, 1)
    b = 255 * yy / max(h - 1, 1)
    img = np.stack([r, g, b], 2)
    img[h // 3:, w // 2:] = 255 - img[h // 3:, w // 2:]
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = {
    # name: (h, w, mode, save kwargs)
    "base444": (29, 37, "RGB", dict(quality=90, subsampling="4:4:4")),
    "base420": (29, 37, "RGB", dict(quality=85, subsampling="4:2:0")),
    "base422": (24, 41, "RGB", dict(quality=75, subsampling="4:2:2")),
    "base411": (19, 45, "RGB", dict(quality=80, subsampling="4:1:1")),
    "lowq420": (40, 56, "RGB", dict(quality=8, subsampling="4:2:0")),
    "q100": (16, 16, "RGB", dict(quality=100, subsampling="4:4:4")),
    "grey": (17, 33, "L", dict(quality=80)),
    "optimised": (31, 30, "RGB", dict(quality=70, subsampling="4:2:0", optimize=True)),
    "prog420": (29, 37, "RGB", dict(quality=85, subsampling="4:2:0", progressive=True)),
    "prog444": (35, 26, "RGB", dict(quality=30, subsampling="4:4:4", progressive=True)),
    "proggrey": (20, 23, "L", dict(quality=60, progressive=True)),
    "restart420": (32, 48, "RGB", dict(quality=85, subsampling="4:2:0", restart_marker_blocks=2)),
    "restartprog": (32, 40, "RGB", dict(quality=60, subsampling="4:2:2", progressive=True, restart_marker_rows=1)),
    "one": (1, 1, "RGB", dict(quality=90, subsampling="4:2:0")),
    "row": (1, 17, "RGB", dict(quality=90, subsampling="4:2:0")),
    "col": (19, 1, "RGB", dict(quality=90, subsampling="4:2:2")),
    "tex64": (64, 64, "RGB", dict(quality=92, subsampling="4:2:0")),
}


# files Pillow cannot write, from tests/jpeg_writer.py: name: (h, w, sampling factors of Y/Cb/Cr, encode kwargs)
OWN = {
    "own440": (33, 21, [(1, 2), (1, 1), (1, 1)], {}),                       # vertical-only chroma subsampling
    "ownmixed": (27, 35, [(2, 2), (1, 2), (2, 1)], {}),                     # Cb halved horizontally, Cr vertically
    "own410": (23, 50, [(4, 2), (1, 1), (1, 1)], dict(restart=1)),          # 4x2: the nearest-neighbour path
    "ownrgbids": (18, 22, [(1, 1), (1, 1), (1, 1)], dict(ids=(82, 71, 66))), # components labelled 'R','G','B': no transform
    "ownadobe0": (18, 22, [(1, 1), (1, 1), (1, 1)], dict(jfif=False, adobe=0)),   # Adobe transform 0 without JFIF: RGB
    "ownadobe0jfif": (18, 22, [(2, 1), (1, 1), (1, 1)], dict(jfif=True, adobe=0)), # ... with JFIF: converted after all
    "ownadobe1": (18, 22, [(1, 1), (1, 1), (1, 1)], dict(jfif=False, adobe=1)),
    "owndqt16": (16, 24, [(2, 2), (1, 1), (1, 1)], dict(dqt16=True, sof=0xc1, quant=[300] + [3] * 63)),
    "ownwrap": (16, 16, [(1, 1), (1, 1), (1, 1)], dict(dqt16=True, sof=0xc1, quant=[40000] + [1] * 63)),  # 16-bit products wrap
}


def own_file(h, w, sampling, kw, seed):
    import jpeg_writer as jw
    img = picture(h, w, seed)
    comps = jw.rgb_to_ycc(img) if kw.get("ids") is None and kw.get("adobe") != 0 else [img[..., k] for k in range(3)]
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    planes = [jw.subsample(c, hmax // s[0], vmax // s[1]) for c, s in zip(comps, sampling)]
    return jw.encode(planes, sampling, w, h, **kw)


def main():
    R = ol.load_ref()
    assert R is not None and hasattr(R, "ref_stbi_load_rgb8"), "build oracle/_ref first: make -C oracle ref"
    out = {}
    files = []
    for i, (name, (h, w, mode, kw)) in enumerate(CASES.items()):
        img = picture(h, w, i + 1)
        pil = Image.fromarray(img if mode == "RGB" else img[..., 1], mode)
        bio = io.BytesIO()
        pil.save(bio, "JPEG", **kw)
        files.append((name, h, w, bio.getvalue()))
    for i, (name, (h, w, sampling, kw)) in enumerate(OWN.items()):
        files.append((name, h, w, own_file(h, w, sampling, kw, 100 + i)))
    for name, h, w, data in files:
        with open(os.path.join(HERE, "jpeg", name + ".jpg"), "wb") as f:
            f.write(data)
        buf = np.frombuffer(data, np.uint8)
        dec = np.zeros((h, w, 3), np.uint8)
        ww, hh, cc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = R.ref_stbi_load_rgb8(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.size), ctypes.c_void_p(dec.ctypes.data),
                                  ctypes.c_size_t(dec.size), ctypes.byref(ww), ctypes.byref(hh), ctypes.byref(cc))
        assert rc == 0 and (ww.value, hh.value) == (w, h), (name, rc)
        out[name] = dec
        out[name + "_channels"] = np.int32(cc.value)
        print(f"{name}: {len(data)} bytes, {w}x{h}, {cc.value} channels")
    np.savez_compressed(os.path.join(HERE, "ref_jpeg.npz"), **out)


if __name__ == "__main__":
    main()
