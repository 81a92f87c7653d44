#!/usr/bin/env python3
"""Time the two dormant estimators (k_shade_alt) on the bench workload: exp_alt.py [frames]   (PT355_LIB selects the build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
for name, which in (("nee", pt.PT_ESTIMATOR_NEE), ("brdf", pt.PT_ESTIMATOR_BRDF)):
    c = pt.PathTracer(1920, 1080)
    c.upload_scene(scene); c.set_camera(pt.default_camera()); c.set_estimator(which)
    seeds = pt.frame_seeds(0, 2 * frames)
    c.render(0, seeds[:frames], 5); c.sync(); c.reset_counters()
    t0 = time.perf_counter(); c.render(frames, seeds[frames:], 5); c.sync(); dt = time.perf_counter() - t0
    k = c.counters(); c.close()
    rays = k["rays_primary"] + k["rays_shadow"] + k["rays_bounce"]
    print(f"{os.environ.get('PT355_LIB', 'default')} {name}: {dt / frames * 1e3:.4f} ms/frame, {rays / dt / 1e6:.0f} Mrays/s, ms_shade {k['ms_shade']:.2f}, ms_trace {k['ms_trace']:.2f}", flush=True)
