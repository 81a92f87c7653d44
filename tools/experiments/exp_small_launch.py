#!/usr/bin/env python3
"""Small launches: ms per frame of pt_render in launch sets of 1..32 frames (contexts of that max_batch) for the trace grid and the
peek flag given in the environment (PT355_EXP_GRID = blocks, PT355_EXP_PEEK = 0/1; experiment builds only).  One process per
setting: the overrides are read once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
scene = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
out = []
for mb in (1, 2, 4, 8, 16, 32):
    c = pt.PathTracer(w, h, max_batch=mb)
    c.upload_scene(scene); c.set_camera(pt.default_camera())
    n = max(64, 4 * mb)
    seeds = pt.frame_seeds(0, 2 * n)
    c.render(0, seeds[:n], 5); c.sync()
    t0 = time.perf_counter()
    c.render(n, seeds[n:], 5); c.sync()
    t1 = time.perf_counter()
    k = c.counters()
    c.close()
    out.append(f"{mb}: {(t1 - t0) / n * 1e3:.3f}")
print(f"{w}x{h} unit=2^{os.environ.get('PT355_EXP_UNIT', 'auto')} peek={os.environ.get('PT355_EXP_PEEK', 'auto')}  ms/frame by frames per launch set: " + "  ".join(out), flush=True)
