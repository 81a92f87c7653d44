#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (closest-hit + NEE shadow rays) of the MIS path on the bunny-class scene at
1920x1080 (BASELINE.json configs[1]).  One step = one frame = one sample per pixel over the whole image.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU; the image is split into 8x8 screen tiles dealt
round-robin to the ranks (scene replicated, same per-frame seeds => pixels identical to 1 GPU); the finished
framebuffer is gathered to rank 0 over RCCL after the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def make_scene(pt, name):
    if name == "bunny":      # configs[1]: 69,451-triangle stand-in mesh, Oren-Nayar, in the Cornell shell
        return pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
    if name == "bunny_ggx":  # configs[2]
        return pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_SPECULAR_METAL, roughness=0.2, spheres=True)
    if name == "cornell":    # configs[0]
        return pt.Scene.cornell()
    raise SystemExit(f"unknown scene {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--trace-dead-paths", action="store_true", help="trace zero-throughput paths like the reference does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=32)
    args = ap.parse_args()

    import numpy as np
    import torch
    import pathtracer_amd as pt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the path has no CPU fallback")

    scene = make_scene(pt, args.scene)
    flags = pt.PT_FLAG_TRACE_DEAD_PATHS if args.trace_dead_paths else 0
    part = pt.PT_PARTITION_TILES if world > 1 else pt.PT_PARTITION_FULL
    ctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, partition=part, rank=rank, world=world)
    ctx.upload_scene(scene)
    cam = pt.default_camera()
    ctx.set_camera(cam)
    seeds = pt.frame_seeds(0, args.warmup + args.steps)

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ctx.render(0, seeds[: args.warmup], args.bounces)
    sync_all()
    ctx.reset_counters()
    sync_all()
    t0 = time.perf_counter()
    ctx.render(args.warmup, seeds[args.warmup:], args.bounces)
    sync_all()
    dt = time.perf_counter() - t0
    cnt = ctx.counters()
    rays = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_bounce"]

    gather_ms = None
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([rays, cnt["paths"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rays_total, paths_total = int(tot[0].item()), int(tot[1].item())
        # framebuffer gather over RCCL/xGMI (outside the timed region, reported separately)
        n_max = max(len(pt.partition_pixels(args.width, args.height, part, r, world)) for r in range(world))
        mine = torch.zeros((n_max, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        ctx.pack_owned(mine.data_ptr())
        ctx.sync()
        bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, bufs, dst=0)
        if rank == 0:
            torch.cuda.synchronize()
            for r in range(1, world):
                ctx.unpack_from(bufs[r].data_ptr(), r, world)
            ctx.sync()
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
    else:
        rays_total, paths_total = rays, cnt["paths"]

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline inputs for the dominant kernel (k_trace): S, T per ray from a counted replay of a few frames
    cctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags | pt.PT_FLAG_COUNT_TRAVERSAL, partition=part,
                         rank=rank, world=world)
    cctx.upload_scene(scene)
    cctx.set_camera(cam)
    nc = min(2, args.steps)
    cctx.render(args.warmup, seeds[args.warmup: args.warmup + nc], args.bounces)
    cc = cctx.counters()
    cctx.close()
    crays = cc["rays_primary"] + cc["rays_shadow"] + cc["rays_bounce"]
    S, T = cc["node_steps"] / crays, cc["tri_tests"] / crays
    bytes_per_ray = 64.0 * S + 64.0 * T + 96.0
    launches = max(cnt["trace_launches"], 1)
    avg_launch_ms = cnt["ms_trace"] / launches
    rays_per_launch = rays / launches
    achieved = bytes_per_ray * rays_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_trace", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "bytes_per_ray": round(bytes_per_ray, 1), "node_steps_per_ray": round(S, 3), "tri_tests_per_ray": round(T, 3),
                "rays_per_launch": round(rays_per_launch, 1), "avg_launch_ms": round(avg_launch_ms, 4),
                "trace_ms_total": round(cnt["ms_trace"], 3), "shade_ms_total": round(cnt["ms_shade"], 3)}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle_lib as ol
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        osc = ol.scene_from_desc(scene.desc())
        nf = max(1, args.cpu_frames)
        c0 = time.perf_counter()
        _, oc = ol.render(osc, cam, args.width, args.height, pt.frame_seeds(args.warmup, nf), first_frame=args.warmup,
                          bounces=args.bounces, threads=threads)
        cdt = time.perf_counter() - c0
        orays = oc["rays_primary"] + oc["rays_shadow"] + oc["rays_bounce"]
        cpu = {"value": round(orays / cdt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": f"{nf} frame(s) of the same {args.width}x{args.height} workload (oracle traces zero-throughput paths too)",
               "spp_per_s": round(nf / cdt, 4), "seconds": round(cdt, 2)}

    out = {
        "metric": "Mrays/s (closest-hit + NEE shadow rays), MIS path, bunny-class scene 1920x1080",
        "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene}: {scene.desc().n_tris} triangles, Sweep-SAH BVH {scene.desc().n_nodes} nodes, "
                               f"{args.width}x{args.height}, {args.steps} spp, {args.bounces} bounces, MIS (BSDF + NEE), "
                               f"shadow rays closest-hit (reference semantics)",
                   "scene_hash": f"{scene.hash():016x}", "partition": "tiles8x8-roundrobin" if world > 1 else "single",
                   "dead_paths": "traced" if args.trace_dead_paths else "retired"},
        "spp_per_s": round(args.steps / dt, 3), "mpaths_per_s": round(paths_total / dt / 1e6, 2),
        "rays": {"primary": cnt["rays_primary"], "shadow": cnt["rays_shadow"], "bounce": cnt["rays_bounce"], "scope": "rank0" if world > 1 else "all"},
        "stack_spills": cnt["stack_spills"], "roofline": roofline, "cpu_baseline": cpu,
    }
    if gather_ms is not None:
        out["gather_ms"] = round(gather_ms, 3)
    print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
