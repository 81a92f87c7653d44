#!/usr/bin/env python3
"""PLOC trees of the bench mesh (69 k and 1 M triangles): build time, S, T and Grays/s (GPU box).  Run with a library built with
another -DPLOC_RADIUS to compare search radii (tools/exp_build.sh + LD path is not needed: the package loads pathtracer_amd/libpt355.so)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt

def run(scene, label):
    out = []
    for flags in (0, pt.PT_FLAG_COUNT_TRAVERSAL):
        c = pt.PathTracer(1920, 1080, flags=flags)
        c.upload_scene(scene); c.set_camera(pt.default_camera())
        seeds = pt.frame_seeds(0, 64)
        c.render(0, seeds[:32], 5); c.sync(); c.reset_counters()
        t0 = time.time(); c.render(32, seeds[32:], 5); c.sync(); dt = time.time() - t0
        k = c.counters(); rays = k["rays_primary"] + k["rays_shadow"] + k["rays_bounce"]
        out.append((rays / dt / 1e6, k["ms_trace"], k["node_steps"] / rays, k["tri_tests"] / rays))
        c.close()
    print(f"{label:40s} {out[0][0]:8.1f} Mrays/s  trace {out[0][1]:6.1f} ms | S {out[1][2]:5.2f} T {out[1][3]:4.2f}", flush=True)

for freq in (59, 224):
    for ml in (2, 4):
        sc = pt.Scene.bunny(freq=freq, target_tris=69451 if freq == 59 else 0, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
        c = pt.PathTracer(8, 8); c.finalize_scene_device(sc, ml, ploc=True); ms = c.finalize_scene_device(sc, ml, ploc=True); c.close()
        run(sc, f"{sys.argv[1] if len(sys.argv) > 1 else ''} freq {freq} PLOC max_leaf {ml} ({ms:6.2f} ms)")
