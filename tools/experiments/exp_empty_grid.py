#!/usr/bin/env python3
"""What do k_shade launches over EMPTY queues cost (GPU box)?  A scene whose two triangles sit behind the camera: every primary ray
misses, the stages after stage 0 find nothing in their queues, yet their grids cover every slot."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt

base = pt.Scene.soup(2, seed=3).arrays()
tris = base["tris"][:2].copy()
tf = tris.view(np.float32)
for i in range(2):
    tf[i, 0:3] = (0.0 + i, 0.0, 1000.0); tf[i, 4:7] = (0.1 + i, 0.0, 1000.0); tf[i, 8:11] = (0.0 + i, 0.1, 1000.0)
nodes = np.zeros((3, 8), np.float32)
nodes[0, 0:3], nodes[0, 4:7], nodes[0, 3] = (0, 0, 999.9), (1.2, 0.2, 1000.1), 1.0
nodes[1, 0:3], nodes[1, 4:7], nodes[1, 3], nodes[1, 7] = (0, 0, 999.9), (0.2, 0.2, 1000.1), 0.0, 1.0
nodes[2, 0:3], nodes[2, 4:7], nodes[2, 3], nodes[2, 7] = (1.0, 0, 999.9), (1.2, 0.2, 1000.1), 1.0, 1.0
d = pt.pt_scene_desc()
keep = []
for field, count, arr in (("spheres", "n_spheres", np.zeros((0, 8), np.uint32)), ("tris", "n_tris", tris), ("light_tris", "n_light_tris", np.zeros(0, np.uint32)),
                          ("materials", "n_materials", base["materials"]), ("nodes", "n_nodes", nodes.view(np.uint32)), ("light_spheres", "n_light_spheres", np.zeros(0, np.uint32))):
    arr = np.ascontiguousarray(arr, np.uint32); keep.append(arr)
    setattr(d, field, arr.ctypes.data if arr.size else None); setattr(d, count, arr.shape[0])
for b in (0, 1, 5):
    c = pt.PathTracer(1920, 1080)
    c.upload_scene(d); c.set_camera(pt.default_camera())
    seeds = pt.frame_seeds(0, 64)
    c.render(0, seeds[:32], b); c.sync(); c.reset_counters()
    c.render(32, seeds[32:], b); c.sync()
    k = c.counters()
    print(f"bounces {b}: ms_trace {k['ms_trace']:.3f} ms_shade {k['ms_shade']:.3f} rays {k['rays_primary'] + k['rays_shadow'] + k['rays_bounce']}", flush=True)
    c.close()
