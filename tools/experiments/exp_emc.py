#!/usr/bin/env python3
"""GPU builders with plain vs extended Morton codes (PT355_BVH_PLAIN_MORTON set / unset) on the bench mesh and on the atrium (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt

def run(scene, label, tex=None):
    out = []
    for flags in (0, pt.PT_FLAG_COUNT_TRAVERSAL):
        c = pt.PathTracer(1920, 1080, flags=flags)
        c.upload_scene(scene, tex_rgb8=tex); c.set_camera(pt.default_camera())
        seeds = pt.frame_seeds(0, 48)
        c.render(0, seeds[:16], 5); c.sync(); c.reset_counters()
        t0 = time.time(); c.render(16, seeds[16:], 5); c.sync(); dt = time.time() - t0
        k = c.counters(); rays = k["rays_primary"] + k["rays_shadow"] + k["rays_bounce"]
        out.append((rays / dt / 1e6, k["ms_trace"], k["node_steps"] / rays, k["tri_tests"] / rays))
        c.close()
    print(f"{label:44s} {out[0][0]:8.1f} Mrays/s  trace {out[0][1]:6.1f} ms | S {out[1][2]:5.2f} T {out[1][3]:4.2f}", flush=True)

for name in ("bunny", "atrium"):
    tex = pt.make_textures(6, 1) if name == "atrium" else None
    mk = (lambda: pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)) if name == "bunny" else (lambda: pt.Scene.atrium(detail=4, seed=1))
    run(mk(), f"{name}: Sweep-SAH", tex)
    for ploc in (False, True):
        for emc in (0, 1):
            os.environ.pop("PT355_BVH_PLAIN_MORTON", None)
            if not emc:
                os.environ["PT355_BVH_PLAIN_MORTON"] = "1"
            sc = mk()
            c = pt.PathTracer(8, 8); ms = c.finalize_scene_device(sc, 2, ploc=ploc); c.close()
            run(sc, f"{name}: {'PLOC' if ploc else 'LBVH'} emc={emc} ({ms:5.2f} ms)", tex)
