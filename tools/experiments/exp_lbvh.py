#!/usr/bin/env python3
"""Bench scene traced through the reference builder's Sweep-SAH tree vs the GPU-built LBVH (GPU box only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt

def run(scene, label, flags=0):
    c = pt.PathTracer(1920, 1080, flags=flags)
    c.upload_scene(scene); c.set_camera(pt.default_camera())
    seeds = pt.frame_seeds(0, 48)
    c.render(0, seeds[:16], 5); c.sync(); c.reset_counters()
    t0 = time.time(); c.render(16, seeds[16:], 5); c.sync(); dt = time.time() - t0
    k = c.counters(); rays = k["rays_primary"] + k["rays_shadow"] + k["rays_bounce"]
    extra = f" S={k['node_steps']/rays:.2f} T={k['tri_tests']/rays:.2f}" if flags else ""
    print(f"{label:28s} {rays/dt/1e6:8.1f} Mrays/s  trace {k['ms_trace']:.1f} ms shade {k['ms_shade']:.1f} ms{extra}", flush=True)
    c.close()

t0 = time.time(); sah = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175); t_cpu = time.time() - t0
print(f"scene generation + CPU Sweep-SAH build: {t_cpu*1e3:.0f} ms")
for fl in (0, 4):
    run(sah, "Sweep-SAH (reference builder)", fl)
for ploc in (False, True):
    name = "PLOC" if ploc else "LBVH"
    for ml in (1, 2, 4, 8):
        lb = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
        c = pt.PathTracer(8, 8); ms = c.finalize_scene_device(lb, ml, ploc=ploc); ms = c.finalize_scene_device(lb, ml, ploc=ploc); c.close()
        print(f"{name} max_leaf {ml}: build {ms:.2f} ms, {lb.arrays()['nodes'].shape[0]} nodes")
        for fl in (0, 4):
            run(lb, f"{name} max_leaf {ml}", fl)
