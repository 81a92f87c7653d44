#!/usr/bin/env python3
"""Writes one of the procedural stand-in scenes as a .glb that the loader (pth_load_gltf = the reference's LoadGLTF)
reads, so that BASELINE.json configs[3]/[4] ("glTF ... with loaded materials") run through the glTF path:

    python tools/make_scene_glb.py atrium out.glb [--detail 4]
    python bench.py --scene out.glb

What the reference's loader can represent decides the shape of the file (src/loader.cpp): uint16 indices -> chunks of at
most 21,845 triangles; ONE primitive per mesh (several primitives per mesh would re-emit each other's triangles,
loader.cpp:304-345); one glTF texture entry per textured primitive (one array layer is filled per textured primitive,
loader.cpp:235); lights through KHR_materials_emissive_strength (loader.cpp:177-185).  Materials come back the way that
loader maps them: rough dielectrics as Oren-Nayar (roughness * 0.35), metals as Lambert with roughness 2.0.  Positions are
written in model space: the inverse of the placement src/main.cpp:26-29 applies (Translate(0,-2,-6) Rotate(-90 deg y) Scale(2))."""
import argparse
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gltf_writer as gw  # noqa: E402
import pathtracer_amd as pt  # noqa: E402


def png_rgb(img):
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].astype(np.uint8).tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 3)) + chunk(b"IEND", b"")


def write_scene_glb(scene, path, textures=None, chunk_tris=21845):
    fat = scene.fat_triangles()
    mats = scene.arrays()["materials"].view(np.float32)   # [m, 12]: diffuse rgb, rough | specular rgb, type | Le rgb, tex
    # world -> model space: p_model = Ry(-90 deg)^-1 (p_world - (0,-2,-6)) / 2; normals rotate the same way
    def to_model(p, is_point):
        q = p.reshape(-1, 3).astype(np.float64)
        if is_point:
            q = q - np.array([0.0, -2.0, -6.0])
        q = np.stack([q[:, 2], q[:, 1], -q[:, 0]], 1)  # the inverse of Rotate(0, -90 deg, 0), which maps (x, y, z) to (-z, y, x)
        if is_point:
            q = q / 2.0
        return q.astype(np.float32)
    v, n, uv = to_model(fat["v"], True).reshape(-1, 3, 3), to_model(fat["n"], False).reshape(-1, 3, 3), fat["uv"].reshape(-1, 3, 2)
    g = gw.Glb()
    images = {}
    g.j["images"], g.j["textures"], g.j["samplers"] = [], [], [{"magFilter": 9729, "minFilter": 9729, "wrapS": 10497, "wrapT": 10497}]
    n_prims = 0
    for m in np.unique(fat["mat"]):
        rec = mats[m]
        ty, tex = int(rec[7]), int(rec[11])
        sel = np.nonzero(fat["mat"] == m)[0]
        for c0 in range(0, sel.size, chunk_tris):
            idx = sel[c0: c0 + chunk_tris]
            if ty == pt.MAT_LIGHT:
                le = rec[8:11].astype(np.float64)
                s = float(le.max())
                gm = g.material(emissive=le / s, strength=s)
            elif ty == pt.MAT_SPECULAR_METAL:
                gm = g.material(base=list(rec[4:7]) + [1.0], metallic=1.0, roughness=float(rec[3]))
            elif ty == pt.MAT_OREN_NAYAR:
                gm = g.material(base=list(rec[0:3]) + [1.0], metallic=0.0, roughness=min(1.0, float(rec[3]) / 0.35))
            else:
                gm = g.material(base=list(rec[0:3]) + [1.0], metallic=0.0, roughness=0.0)
            if tex >= 0 and textures is not None and ty != pt.MAT_LIGHT:
                if tex not in images:
                    images[tex] = len(g.j["images"])
                    g.j["images"].append({"bufferView": g.view(png_rgb(textures[tex])), "mimeType": "image/png"})
                g.j["textures"].append({"source": images[tex], "sampler": 0})
                g.j["materials"][gm]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": len(g.j["textures"]) - 1}
            k = idx.size * 3
            prim = g.primitive(v[idx].reshape(k, 3), n[idx].reshape(k, 3), np.arange(k, dtype=np.uint16), uv[idx].reshape(k, 2), gm)
            mesh = g.mesh([prim])
            if n_prims == 0:
                g.node(mesh=mesh)   # one node is enough: the loader bakes every node's transform into ALL triangles (loader.cpp:349-393)
            n_prims += 1
    for k in ("images", "textures"):
        if not g.j[k]:
            del g.j[k]
    if "textures" not in g.j:
        del g.j["samplers"]
    with open(path, "wb") as f:
        f.write(g.tobytes())
    return n_prims


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", choices=["atrium", "bunny", "cornell"])
    ap.add_argument("out")
    ap.add_argument("--detail", type=int, default=4)
    args = ap.parse_args()
    if args.scene == "atrium":
        sc, tex = pt.Scene.atrium(detail=args.detail, seed=1), pt.make_textures(6, 1)
    elif args.scene == "bunny":
        sc, tex = pt.Scene.bunny(), None
    else:
        sc, tex = pt.Scene.cornell(), None
    n = write_scene_glb(sc, args.out, tex)
    print(f"{args.out}: {sc.fat_triangles()['mat'].size} triangles in {n} single-primitive meshes, {os.path.getsize(args.out) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
