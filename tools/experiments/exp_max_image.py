#!/usr/bin/env python3
"""Maximum image size pt_create accepts (16384 x 16384 = 2^28 paths per frame: path ids fill the 28 bits k_shade packs them into):
one MIS frame of the Cornell box with spheres, HIP vs oracle, every pixel.   exp_max_image.py [side]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pathtracer_amd as pt, oracle_lib as ol
side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
scene = pt.Scene.cornell(spheres=True)
cam, seeds = pt.default_camera(), pt.frame_seeds(5, 1)
t0 = time.perf_counter()
c = pt.PathTracer(side, side, flags=pt.PT_FLAG_TRACE_DEAD_PATHS)
c.upload_scene(scene); c.set_camera(cam)
c.render(5, seeds, 5); c.sync()
t1 = time.perf_counter()
img = c.read_framebuffer(); cnt = c.counters(); info = [int(v) for v in c.info]; c.close()
print(f"{side}x{side}: GPU {t1 - t0:.2f} s incl. set-up, frames per launch set {info[0]}, queue stride {info[2]}, rays {cnt['rays_primary'] + cnt['rays_shadow'] + cnt['rays_bounce']}", flush=True)
ref = np.zeros((side, side, 4), np.float32)
t2 = time.perf_counter()
ref, ocnt = ol.render(ol.scene_from_desc(scene.desc()), cam, side, side, seeds, first_frame=5, threads=16, rgba=ref)
print(f"oracle {time.perf_counter() - t2:.1f} s", flush=True)
bad = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
same_counts = all(cnt[k] == ocnt[k] for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce"))
print(f"differing pixels: {bad} of {side * side}; ray counts equal: {same_counts}")
sys.exit(0 if bad == 0 and same_counts else 1)
