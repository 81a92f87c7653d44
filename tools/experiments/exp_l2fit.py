#!/usr/bin/env python3
"""Does a traversal blob that fits the XCD's 4 MiB L2 trace faster per step? (GPU box)  The bench mesh at 9 k … 70 k
triangles (0.9 … 7 MB of blob), 1080p MIS, Sweep-SAH tree: trace time per (node-pair step + triangle test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt

for freq in [int(x) for x in sys.argv[1:]] or (21, 30, 36, 42, 48, 53, 59, 70, 84):
    sc = pt.Scene.bunny(freq=freq, target_tris=0, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
    res = []
    for flags in (0, pt.PT_FLAG_COUNT_TRAVERSAL):
        c = pt.PathTracer(1920, 1080, flags=flags)
        c.upload_scene(sc); c.set_camera(pt.default_camera())
        seeds = pt.frame_seeds(0, 64)
        c.render(0, seeds[:32], 5); c.sync(); c.reset_counters()
        c.render(32, seeds[32:], 5); c.sync()
        res.append(c.counters()); c.close()
    k, kc = res
    rays = kc["rays_primary"] + kc["rays_shadow"] + kc["rays_bounce"]
    a = sc.arrays()
    mb = (a["nodes"].shape[0] // 2 * 64 + a["tris"].shape[0] * 64) / 1e6
    steps = kc["node_steps"] + kc["tri_tests"]
    print(f"freq {freq:3d} {a['tris'].shape[0]:7d} tris {mb:5.2f} MB | trace {k['ms_trace']:6.2f} ms shade {k['ms_shade']:6.2f} ms | "
          f"S {kc['node_steps']/rays:5.2f} T {kc['tri_tests']/rays:4.2f} | {steps / k['ms_trace'] / 1e6:6.1f} G lane-steps/s | "
          f"top share {kc['top_steps']/max(1, steps):.3f} wave-iters {kc['wave_iters']/1e6:.1f} M", flush=True)
