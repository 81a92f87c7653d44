#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/.  Run in the BUILD CONTAINER only (it needs /root/reference):

    python tests/golden/make_golden.py

Two families of vectors, kept in separate files because they are pinned by different things:

  ref_host.npz     produced by the REFERENCE'S OWN translation units (src/scene/{bvh,triangle,material,
                   sphere}.cpp, src/math/math.cpp, thirdparty/pcg-c-basic-0.9/pcg_basic.c) compiled from
                   /root/reference into oracle/_ref/libptref.so by oracle/Makefile.  Inputs are synthetic and
                   seeded; outputs are the reference's bytes.  These pin the host half of the path: SSBO
                   packing, the Sweep-SAH BVH build, emissive lists and the per-frame host PCG seed.

  oracle_*.npz     produced by oracle/liboracle.so, the build's CPU restatement of shaders/framebuffer.comp.
                   The reference ships no golden vectors for the integrator and its GLSL cannot run offline
                   (SURVEY.md 8c), so these are regression pins of the restatement ("parity unpinned" by the
                   reference itself); the HIP path must reproduce them bit for bit.

Fixtures are data only (inputs + expected outputs); nothing of the reference's source text is stored.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
import pathtracer_amd as pt  # noqa: E402
import scene_textured_spheres as sts  # noqa: E402
from golden_inputs import fat_triangles, digest64, lcg_mesh, materials_in, spheres_in  # noqa: E402

P = ctypes.c_void_p


def ptr(a):
    return a.ctypes.data_as(P)


def main():
    R = ol.load_ref()
    assert R is not None, "needs /root/reference (build container)"
    out = {}

    # ---- layout sizes and the host PCG stream (src/main.cpp:97)
    sizes = np.zeros(4, np.uint32)
    R.ref_sizes(ptr(sizes))
    out["sizes_tri_node_mat_sphere"] = sizes
    R.ref_pcg32_restart()
    out["pcg_seeds"] = np.array([R.ref_pcg32_next() for _ in range(256)], np.uint32)

    # ---- packing (TriangleGLSL / MaterialGLSL / SphereGLSL ctors, octahedral encoding)
    v, uv, n, mat = fat_triangles(512, seed=3)
    packed = np.zeros((512, 16), np.uint32)
    for i in range(512):
        R.ref_pack_triangle(ptr(v[i]), ptr(uv[i]), ptr(n[i]), ctypes.c_uint32(int(mat[i])), ptr(packed[i]))
    out["tri_packed"] = packed
    vecs = np.concatenate([n.reshape(-1, 3), np.eye(3, dtype=np.float32), -np.eye(3, dtype=np.float32)])
    enc = np.zeros((vecs.shape[0], 2), np.float32)
    for i in range(vecs.shape[0]):
        R.ref_oct_encode(ptr(vecs[i]), ptr(enc[i]))
    out["oct_in"], out["oct_out"] = vecs, enc
    mi = materials_in()
    mp = np.zeros((len(mi), 12), np.uint32)
    for i, (d, s, le, rough, tex, ty) in enumerate(mi):
        R.ref_pack_material(ptr(d), ptr(s), ptr(le), ctypes.c_float(rough), ctypes.c_int(tex), ctypes.c_int(ty), ptr(mp[i]))
    out["mat_packed"] = mp
    si = spheres_in()
    sp = np.zeros((len(si), 8), np.uint32)
    for i, (o, r, m) in enumerate(si):
        R.ref_pack_sphere(ptr(o), ctypes.c_float(r), ctypes.c_uint32(m), ptr(sp[i]))
    out["sphere_packed"] = sp

    # ---- CalculateBVH: full bytes for small meshes, SHA-256/64(nodes || sorted) for large ones
    R.ref_calculate_bvh.restype = ctypes.c_uint32
    sizes_full = [2, 3, 5, 16, 17, 33, 64, 257, 1000]
    sizes_hash = [5000, 20000, 200000]
    for nt in sizes_full + sizes_hash:
        tris = lcg_mesh(nt, seed=nt)
        nodes = np.zeros((2 * nt + 1, 8), np.uint32)
        srt = np.zeros((nt, 16), np.uint32)
        nn = R.ref_calculate_bvh(ptr(tris), nt, ptr(nodes), ptr(srt))
        nodes = nodes[:nn]
        if nt in sizes_full:
            out[f"bvh_{nt}_nodes"], out[f"bvh_{nt}_sorted"] = nodes, srt
        out[f"bvh_{nt}_hash"] = np.array([nn, digest64(nodes.tobytes() + srt.tobytes())], np.uint64)
    # clustered input: many equal centroids / degenerate boxes (stresses the split tie-breaks)
    tris = lcg_mesh(3000, seed=99, snap=0.25)
    nodes = np.zeros((6001, 8), np.uint32)
    srt = np.zeros((3000, 16), np.uint32)
    nn = R.ref_calculate_bvh(ptr(tris), 3000, ptr(nodes), ptr(srt))
    out["bvh_snap3000_hash"] = np.array([nn, digest64(nodes[:nn].tobytes() + srt.tobytes())], np.uint64)

    # ---- the build's own procedural scenes, re-built by the reference builder from the same triangle set
    for name, sc in (("cornell", pt.Scene.cornell(spheres=True)), ("bunny12", pt.Scene.bunny(freq=12, target_tris=0, spheres=True)),
                     ("soup3000", pt.Scene.soup(3000, seed=5))):
        a = sc.arrays()
        nt = a["tris"].shape[0]
        nodes = np.zeros((2 * nt + 1, 8), np.uint32)
        srt = np.zeros((nt, 16), np.uint32)
        nn = R.ref_calculate_bvh(ptr(a["tris"]), nt, ptr(nodes), ptr(srt))
        out[f"scene_{name}_rebuild_hash"] = np.array([nn, digest64(nodes[:nn].tobytes() + srt.tobytes())], np.uint64)
        lt = np.zeros(max(nt, 1), np.uint32)
        R.ref_find_emissive_tris.restype = ctypes.c_uint32
        k = R.ref_find_emissive_tris(ptr(a["tris"]), nt, ptr(a["materials"]), a["materials"].shape[0], ptr(lt))
        out[f"scene_{name}_light_tris"] = lt[:k]
        ls = np.zeros(max(a["spheres"].shape[0], 1), np.uint32)
        R.ref_find_emissive_spheres.restype = ctypes.c_uint32
        k = R.ref_find_emissive_spheres(ptr(a["spheres"]), a["spheres"].shape[0], ptr(a["materials"]), a["materials"].shape[0], ptr(ls))
        out[f"scene_{name}_light_spheres"] = ls[:k]
    # emissive rules: ANY component for triangles, ALL components for spheres (sphere.cpp:17)
    le_cases = np.array([[0, 0, 0], [1e-4, 0, 0], [0.99e-4, 0, 0], [0, 5, 0], [1, 1, 1], [1, 1, 0], [1e-4, 1e-4, 1e-4], [2e-4, 2e-4, 2e-4]], np.float32)
    mats = np.zeros((len(le_cases), 12), np.uint32)
    z = np.zeros(3, np.float32)
    for i, le in enumerate(le_cases):
        R.ref_pack_material(ptr(z), ptr(z), ptr(le), ctypes.c_float(0.0), ctypes.c_int(-1), ctypes.c_int(-1), ptr(mats[i]))
    tri8 = packed[: len(le_cases)].copy()
    tri8[:, 15] = np.arange(len(le_cases))
    sph8 = np.zeros((len(le_cases), 8), np.uint32)
    for i in range(len(le_cases)):
        R.ref_pack_sphere(ptr(z), ctypes.c_float(1.0), ctypes.c_uint32(i), ptr(sph8[i]))
    lt = np.zeros(16, np.uint32)
    k = R.ref_find_emissive_tris(ptr(tri8), len(le_cases), ptr(mats), len(le_cases), ptr(lt))
    out["emissive_rule_mats"], out["emissive_rule_tris"] = mats, lt[:k].copy()
    k = R.ref_find_emissive_spheres(ptr(sph8), len(le_cases), ptr(mats), len(le_cases), ptr(lt))
    out["emissive_rule_spheres"] = lt[:k].copy()
    np.savez_compressed(os.path.join(HERE, "ref_host.npz"), **out)
    print("ref_host.npz:", len(out), "arrays")

    # ---- oracle: closest-hit batches
    tr = {}
    for name, sc in (("cornell_spheres", pt.Scene.cornell(spheres=True)), ("soup3000", pt.Scene.soup(3000, seed=5)),
                     ("bunny12", pt.Scene.bunny(freq=12, target_tris=0, spheres=True)), ("textured_spheres", sts.build(pt)[0])):
        ro, rd = golden_rays(4096, seed=len(name))
        prim, th, kind, uvv, steps, tests = ol.trace_closest(ol.scene_from_desc(sc.desc()), ro, rd, counts=True)
        tr[f"{name}_hash"] = np.array([sc.hash()], np.uint64)
        tr[f"{name}_ro"], tr[f"{name}_rd"] = ro, rd
        tr[f"{name}_prim"], tr[f"{name}_th"], tr[f"{name}_kind"], tr[f"{name}_uv"] = prim, th, kind, uvv
        tr[f"{name}_steps"], tr[f"{name}_tests"] = steps.astype(np.uint16), tests.astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "oracle_trace.npz"), **tr)
    print("oracle_trace.npz:", len(tr), "arrays")

    # ---- oracle: small MIS frames (64x64, 4 spp, 5 bounces, host-PCG seeds 0..3)
    fr = {}
    cam = pt.default_camera()
    seeds = pt.frame_seeds(0, 4)
    for name, sc, env in golden_frame_scenes(pt):
        tex = sts.build(pt)[1] if name == "textured_spheres" else None
        img, cnt = ol.render(ol.scene_from_desc(sc.desc(), env_rgb8=env, tex_rgb8=tex), cam, 64, 64, seeds, bounces=5)
        fr[f"{name}_hash"] = np.array([sc.hash()], np.uint64)
        fr[f"{name}_rgba"] = img
        fr[f"{name}_counts"] = np.array([cnt[k] for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce", "node_steps", "tri_tests")], np.uint64)
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **fr)
    print("oracle_frames.npz:", len(fr), "arrays")


def golden_rays(n, seed):
    rng = np.random.default_rng(seed)
    ro = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    ro[: n // 4] = 0.0  # camera position
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[: n // 4, 2] = -np.abs(rd[: n // 4, 2]) - 1.0
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[-64:-32, 0] = 0.0   # inv_dir = inf
    rd[-32:] = [0, -1, 0]
    return ro, rd.astype(np.float32)


def golden_env():
    y, x = np.mgrid[0:32, 0:64]
    return np.stack([(x * 4) % 256, (y * 8) % 256, (x * y) % 256], 2).astype(np.uint8)


def golden_frame_scenes(pt):
    return [("cornell", pt.Scene.cornell(), None),
            ("cornell_spheres", pt.Scene.cornell(spheres=True), None),
            ("bunny12_on", pt.Scene.bunny(freq=12, target_tris=0), None),
            ("bunny12_ggx_env", pt.Scene.bunny(freq=12, target_tris=0, material=pt.MAT_SPECULAR_METAL, roughness=0.2, spheres=True), golden_env()),
            ("textured_spheres", sts.build(pt)[0], None)]


if __name__ == "__main__":
    main()
