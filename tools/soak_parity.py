#!/usr/bin/env python3
"""Randomised parity soak (GPU box): many small random configurations -- scene family and size, materials, spheres, textures,
camera pose, image size, frame count, bounce count, batch size, GPU-built trees -- each rendered by the HIP path and by
the oracle, compared bit for bit (image, ray counts).  Not part of the test suite (it runs as long as it is given);
prints one line per mismatch and a summary.    usage: soak_parity.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
import oracle_lib as ol
import scene_textured_spheres as sts

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def camera():
    o = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(-7.5, -1.0)], np.float32)
    if rng.random() < 0.3:
        o[:] = 0
    f = rng.normal(size=3).astype(np.float32)
    if rng.random() < 0.5:
        f = np.array([0, 0, -1], np.float32) + 0.3 * f
    f /= np.linalg.norm(f)
    up = np.array([0, 1, 0], np.float32)
    r = np.cross(f, up)
    if np.linalg.norm(r) < 1e-3:
        r = np.array([1, 0, 0], np.float32)
    r = (r / np.linalg.norm(r)).astype(np.float32)
    cam = np.zeros(12, np.float32)
    pt.lib().pth_pack_camera(o.ctypes.data, f.ctypes.data, r.ctypes.data, 0.005, 0.05, cam.ctypes.data)
    return cam


def scene():
    k = rng.integers(0, 6)
    tex = None
    if k == 0:
        sc, name = pt.Scene.cornell(spheres=bool(rng.integers(0, 2))), "cornell"
    elif k == 1:
        n = int(rng.choice([1, 2, 3, 7, 50, 400, 3000]))
        sc, name = pt.Scene.soup(n, seed=int(rng.integers(1, 1 << 30)), size=float(rng.choice([0.02, 0.25, 1.0, 3.0]))), f"soup{n}"
    elif k == 2 or k == 3:
        mat = int(rng.choice([pt.MAT_LAMBERTIAN, pt.MAT_OREN_NAYAR, pt.MAT_SPECULAR_METAL]))
        rough = float(rng.choice([0.0, 0.01, 0.1, 0.175, 0.5, 1.0]))
        sc = pt.Scene.bunny(freq=int(rng.integers(3, 20)), target_tris=0, seed=int(rng.integers(1, 1 << 30)), material=mat, roughness=rough,
                            spheres=bool(rng.integers(0, 2)))
        name = f"bunny(m{mat},r{rough})"
    elif k == 4:
        sc, name = pt.Scene.atrium(detail=1, seed=int(rng.integers(1, 1 << 30))), "atrium"
        tex = pt.make_textures(6, int(rng.integers(1, 100)))
    else:
        # spheres with TEXTURED materials in front of triangles with uvs: a sphere hit keeps the last accepted triangle's uvs (fc:447-451)
        extra = [((rng.uniform(-1.6, 1.6), rng.uniform(-1.6, 1.6), rng.uniform(-7.6, -4.4)), rng.uniform(0.05, 0.6),
                  int(rng.choice([pt.MAT_LAMBERTIAN, pt.MAT_OREN_NAYAR, pt.MAT_SPECULAR_METAL])), float(rng.choice([0.0, 0.05, 0.2, 0.6])),
                  int(rng.integers(-1, 3))) for _ in range(int(rng.integers(0, 6)))]
        sc, tex = sts.build(pt, wall_uv_shift=float(rng.uniform(-2, 2)), extra=extra)
        name = f"textured_spheres(+{len(extra)})"
    return sc, tex, name


t_end = time.time() + budget
n = bad = lit = 0
rays = 0
while time.time() < t_end:
    sc, tex, name = scene()
    w, h = int(rng.choice([8, 64, 96, 128])), int(rng.choice([8, 64, 72]))
    frames, bounces = int(rng.integers(1, 4)), int(rng.choice([0, 1, 2, 5, 5, 5, 6]))
    first = int(rng.integers(0, 1000))
    seeds = pt.frame_seeds(first, frames)
    cam = camera()
    gpu_tree = rng.random() < 0.25 and sc.desc().n_tris >= 2
    ploc = bool(rng.integers(0, 2))
    est, oflag = ([(pt.PT_ESTIMATOR_MIS, 0)] * 4 + [(pt.PT_ESTIMATOR_NEE, ol.FLAG_EST_NEE), (pt.PT_ESTIMATOR_BRDF, ol.FLAG_EST_BRDF)])[int(rng.integers(0, 6))]
    gflags = pt.PT_FLAG_TRACE_DEAD_PATHS | (pt.PT_FLAG_SHADOW_FULL_TRAVERSAL if rng.random() < 0.3 else 0)
    c = pt.PathTracer(w, h, flags=gflags, max_batch=int(rng.choice([0, 1, 2])), pipelines=int(rng.choice([1, 1, 2])))
    try:
        c.set_estimator(est)
        if gpu_tree:
            c.finalize_scene_device(sc, int(rng.choice([1, 2, 4, 16])), ploc=ploc)
        c.upload_scene(sc, tex_rgb8=tex)
        c.set_camera(cam)
        c.render(first, seeds, bounces)
        img, cnt = c.read_framebuffer(), c.counters()
    finally:
        c.close()
    ref, ocnt = ol.render(ol.scene_from_desc(sc.desc(), tex_rgb8=tex), cam, w, h, seeds, first_frame=first, bounces=bounces, threads=16, flags=oflag)
    diff = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    cdiff = [k for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce") if cnt[k] != ocnt[k]]
    n += 1
    lit += int((ref[..., :3] != 0).any())
    rays += cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_bounce"]
    if n % 500 == 0:
        print(f"... {n} configurations ({lit} with a lit image, {rays / 1e6:.1f} M rays), {bad} mismatches, {t_end - time.time():.0f} s left", flush=True)
    if diff or cdiff:
        bad += 1
        print(f"MISMATCH #{n}: {name} {w}x{h} frames {frames} first {first} bounces {bounces} gpu_tree {gpu_tree}/{ploc} est {est} flags {gflags}: {diff} pixels, counters {cdiff}", flush=True)
print(f"soak: {n} configurations ({lit} with a lit image, {rays / 1e6:.1f} M rays), {bad} mismatches, {budget:.0f} s budget")
sys.exit(1 if bad else 0)
