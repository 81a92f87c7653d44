#!/usr/bin/env python3
"""Renders a generator scene on the GPU and writes the tonemapped image as PNG (stdlib zlib writer).
   usage: render_png.py <atrium|bunny|cornell> <out.png> [width height spp]"""
import os, struct, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


name, out = sys.argv[1], sys.argv[2]
w, h, spp = (int(x) for x in sys.argv[3:6]) if len(sys.argv) >= 6 else (640, 360, 64)
tex = None
if name == "atrium":
    scene, tex = pt.Scene.atrium(detail=4), pt.make_textures(6, 1)
elif name == "bunny":
    scene = pt.Scene.bunny()
else:
    scene = pt.Scene.cornell(spheres=True)
c = pt.PathTracer(w, h)
c.upload_scene(scene, tex_rgb8=tex)
c.set_camera(pt.default_camera())
c.render(0, pt.frame_seeds(0, spp), 5)
img = c.read_framebuffer()
k = c.counters()
print(name, scene.arrays()["tris"].shape[0], "tris;", (k["rays_primary"] + k["rays_shadow"] + k["rays_bounce"]) / 1e6, "Mrays; nan pixels", int(np.isnan(img).any(axis=2).sum()))
write_png(out, pt.tonemap_rgb8(img))
