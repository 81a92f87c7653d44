#!/usr/bin/env python3
"""Behaviour at the limit of float-encoded node indices (BVHNodeGLSL stores child / primitive indices as float, exact below
2^24 = 16,777,216): a tree just below it, and one past it."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, pathtracer_amd as pt
for n in (9_000_000, 10_500_000):
    t0 = time.perf_counter()
    s = pt.Scene.soup(n, seed=3, size=0.02)
    d = s.desc()
    print(n, "triangles ->", d.n_nodes, "nodes, built in %.1f s" % (time.perf_counter() - t0), flush=True)
    c = pt.PathTracer(64, 64)
    try:
        c.upload_scene(s)
        c.set_camera(pt.default_camera())
        c.render(0, pt.frame_seeds(0, 1), 2)
        img = c.read_framebuffer()
        print("   uploaded and rendered; finite:", bool(np.isfinite(img).all()), flush=True)
    except pt.PtError as e:
        print("   refused:", str(e)[:200], flush=True)
    finally:
        c.close()
