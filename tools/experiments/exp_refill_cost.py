#!/usr/bin/env python3
"""What does the refill path of k_trace cost per ray (GPU box)?  Rays that miss the root's box retire after ONE iteration, so a
launch over them is refill + one node step + flush."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt

sc = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
c = pt.PathTracer(8, 8)
c.upload_scene(sc)
n = 1 << 24
rng = np.random.default_rng(1)
ro = np.tile(np.array([[0.0, 0.0, 50.0]], np.float32), (n, 1))      # outside the shell, looking away
rd = rng.normal(size=(n, 3)).astype(np.float32); rd[:, 2] = np.abs(rd[:, 2]) + 0.1
rd /= np.linalg.norm(rd, axis=1, keepdims=True)
for rep in range(3):
    c.reset_counters()
    prim, th, kind, uv = c.trace_closest(ro, rd)
    k = c.counters()
    print(f"{n} rays that miss everything: k_trace {k['ms_trace']:.3f} ms = {k['ms_trace'] * 1e3 / (n / 1e6):.1f} us per million rays; hits {int((kind != 0).sum())}", flush=True)
c.close()
