#!/usr/bin/env python3
"""Trace throughput vs scene size (GPU box): the bench mesh at 70 k, 264 k, 1 M and 4 M triangles, 1080p MIS, Sweep-SAH
tree from the CPU builder, LBVH and PLOC trees from the GPU builders."""
import os, sys, time
import numpy as np
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
        out.append((rays / dt / 1e6, k["ms_trace"], k["ms_shade"], k["node_steps"] / rays, k["tri_tests"] / rays))
        c.close()
    a = scene.arrays()
    mb = (a["nodes"].shape[0] // 2 * 64 + a["tris"].shape[0] * 64) / 1e6
    print(f"{label:34s} {a['tris'].shape[0]:8d} tris {mb:7.1f} MB blob | {out[0][0]:7.1f} Mrays/s  trace {out[0][1]:6.1f} ms  shade {out[0][2]:5.1f} ms | S {out[1][3]:5.2f} T {out[1][4]:4.2f}", flush=True)

for freq in (59, 115, 224, 450):
    t0 = time.time()
    sc = pt.Scene.bunny(freq=freq, target_tris=69451 if freq == 59 else 0, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
    t_cpu = time.time() - t0
    run(sc, f"freq {freq}: Sweep-SAH ({t_cpu:5.1f} s host)")
    c = pt.PathTracer(8, 8)
    t0 = time.time(); ms = c.finalize_scene_device(sc, 2); t_gpu = time.time() - t0
    c.close()
    run(sc, f"freq {freq}: LBVH ({ms:6.2f} ms dev, {t_gpu:4.1f} s)")
    c = pt.PathTracer(8, 8)
    t0 = time.time(); ms = c.finalize_scene_device(sc, 2, ploc=True); t_gpu = time.time() - t0
    c.close()
    run(sc, f"freq {freq}: PLOC ({ms:6.2f} ms dev, {t_gpu:4.1f} s)")
