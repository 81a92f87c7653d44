#!/usr/bin/env python3
"""The reference's own call pattern: ONE dispatch per displayed frame (src/main.cpp:97-101).  Cost of pt_render_frame at the
reference's default 1280x720 (src/defines.hpp:15-16) and at 1080p, back to back and with a sync per frame (a display loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
scene = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
for (w, h) in ((1280, 720), (1920, 1080)):
    c = pt.PathTracer(w, h)
    c.upload_scene(scene); c.set_camera(pt.default_camera())
    seeds = pt.frame_seeds(0, 400)
    for f in range(20): c.render_frame(int(seeds[f]), f, 5)
    c.sync()
    t0 = time.perf_counter()
    for f in range(20, 220): c.render_frame(int(seeds[f]), f, 5)
    c.sync(); t1 = time.perf_counter()
    for f in range(220, 400):
        c.render_frame(int(seeds[f]), f, 5); c.sync()
    t2 = time.perf_counter()
    k = c.counters()
    t3 = time.perf_counter()
    c.render(400, pt.frame_seeds(400, 64), 5); c.sync()
    t4 = time.perf_counter()
    c.close()
    print(f"{w}x{h}: pt_render_frame back to back {(t1 - t0) / 200 * 1e3:.3f} ms/frame, with a sync per frame {(t2 - t1) / 180 * 1e3:.3f} ms/frame, "
          f"pt_render(64 frames) {(t4 - t3) / 64 * 1e3:.3f} ms/frame; kernel ms per frame (events): trace {k['ms_trace'] / 400:.3f} shade {k['ms_shade'] / 400:.3f}", flush=True)
