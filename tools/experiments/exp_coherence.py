#!/usr/bin/env python3
"""How much would ray ordering buy k_trace?  Takes the bounce-1 rays of one 1080p frame of the bench scene out of the
path queue and traces the same set in several orders through pt_trace_closest (GPU box only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt  # noqa: E402


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    w, h = 1920, 1080
    scene = pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
    ctx = pt.PathTracer(w, h, max_batch=1)
    ctx.upload_scene(scene)
    ctx.set_camera(pt.default_camera())
    ctx.render(0, pt.frame_seeds(0, 1), bounces=1)
    ctx.sync()
    rays, cnt = ctx.debug_read_queue(1, 5)
    rays = rays.view(np.float32)
    ro, rd = rays[0:3].T.copy(), rays[3:6].T.copy()
    ok = np.abs(np.linalg.norm(rd, axis=1) - 1.0) < 1e-3
    ro, rd = ro[ok], rd[ok]
    print("bounce-1 rays:", ro.shape[0], "queue count", cnt, flush=True)

    def run(name, order, reps=4):
        o = np.tile(ro[order], (reps, 1))
        d = np.tile(rd[order], (reps, 1))
        ctx.trace_closest(o, d)  # warm
        ctx.reset_counters()
        prim, t_h, kind, uv = ctx.trace_closest(o, d)
        c = ctx.counters()
        n = o.shape[0]
        print(f"{name:28s} {n/1e6:6.2f} Mrays  {c['ms_trace']:8.3f} ms  {n/c['ms_trace']/1e3:8.1f} Mrays/s", flush=True)
        return prim[: ro.shape[0]]

    n = ro.shape[0]
    nat = np.arange(n)
    p0 = run("queue order", nat)
    rng = np.random.default_rng(1)
    perm = rng.permutation(n)
    p1 = run("shuffled", perm)
    assert np.array_equal(p0[perm], p1)
    octant = (rd[:, 0] < 0).astype(np.int64) | ((rd[:, 1] < 0).astype(np.int64) << 1) | ((rd[:, 2] < 0).astype(np.int64) << 2)
    run("octant (stable)", np.argsort(octant, kind="stable"))
    # octant inside chunks of 4096 queue slots (what a per-block binning with 64-ray units could give)
    chunk = nat // 4096
    run("octant inside 4096-chunks", np.lexsort((octant, chunk)))
    chunk = nat // 512
    run("octant inside 512-chunks", np.lexsort((octant, chunk)))
    lo, hi = ro.min(0), ro.max(0)
    q = ((ro - lo) / (hi - lo + 1e-9) * 1023).astype(np.int64)
    m = morton3(q)
    run("morton(origin)", np.argsort(m, kind="stable"))
    run("octant, morton(origin)", np.lexsort((m, octant)))
    m5 = morton3(q >> 5)
    run("morton(origin/32), octant", np.lexsort((octant, m5)))
    # direction-major: quantised direction then origin
    dq = ((rd * 0.5 + 0.5) * 7.999).astype(np.int64)
    dk = dq[:, 0] | (dq[:, 1] << 3) | (dq[:, 2] << 6)
    run("dir 8x8x8, morton(origin)", np.lexsort((m, dk)))


if __name__ == "__main__":
    main()
