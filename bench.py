#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (closest-hit + NEE shadow rays) of the MIS path on the bunny-class scene at
1920x1080 (BASELINE.json configs[1]).  One step = one frame = one sample per pixel over the whole image, per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run, one rank per GPU.  The path shards with no data-path collective: the
scene is replicated and the FRAMES are dealt round-robin (global frame f -> rank f % N, "sample-interleaved",
SURVEY.md 8e), every rank rendering K full-resolution frames with the host-PCG seeds of its own global frame
numbers -- per-GPU work is fixed as N grows (weak scaling), N*K samples per pixel are produced in all.  The only
exchange is the finished framebuffer: one RCCL sum-reduce of (n_r / N_total) * mean_r to rank 0, inside the timed
region.  Rank 0 prints ONE JSON line.
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
    if name == "atrium":     # configs[3] stand-in: 281,620 triangles, 28 materials, texture layers 0..5
        return pt.Scene.atrium(detail=4, seed=1)
    if name == "cornell":    # configs[0]
        return pt.Scene.cornell()
    raise SystemExit(f"unknown scene {name}")


def pmc_traffic(workload_key):
    """HBM-side bytes per k_trace launch from the committed rocprofv3 --pmc passes of this same workload
    (profiles/*_pmc_traffic.json, written by tools/prof_summary.py): bench.py cannot run the profiler on itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload_key and "k_trace" in d.get("kernels", {}):
            best = d
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--max-batch", type=int, default=0, help="frames traced per launch set (0 = library default)")
    ap.add_argument("--trace-dead-paths", action="store_true", help="trace zero-throughput paths like the reference does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=48)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; what the driver runs) or gloo (rehearsal of the N > 1 "
                    "code path with several ranks sharing one GPU: the reduce then goes through host memory)")
    ap.add_argument("--check", action="store_true", help="N > 1: compare the reduced image with a sequential render of the same "
                    "N*K global frames on rank 0 (tolerance 2e-6 relative)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pathtracer_amd as pt
    from pathtracer_amd import multigpu as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank %= max(torch.cuda.device_count(), 1)  # rehearsal: more ranks than GPUs share a device
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the path has no CPU fallback")

    t_scene = time.perf_counter()
    scene = make_scene(pt, args.scene)  # generator + the reference's single-threaded Sweep-SAH build (CalculateBVH)
    t_scene = time.perf_counter() - t_scene
    flags = pt.PT_FLAG_TRACE_DEAD_PATHS if args.trace_dead_paths else 0
    ctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, partition=pt.PT_PARTITION_FULL, max_batch=args.max_batch)
    tex = pt.make_textures(6, 1) if args.scene == "atrium" else None
    t_upload = time.perf_counter()
    ctx.upload_scene(scene, tex_rgb8=tex)
    ctx.sync()
    t_upload = time.perf_counter() - t_upload
    cam = pt.default_camera()
    ctx.set_camera(cam)
    # one seed table for the whole job, indexed by GLOBAL frame number: rank r renders frames r, r + world, ...
    n_global = (args.warmup + args.steps) * world
    seeds = pt.frame_seeds(0, n_global)
    mine = seeds[rank::world]
    n_pix = args.width * args.height
    red = torch.zeros((n_pix, 4), dtype=torch.float32, device="cuda") if dist is not None else None

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def exchange(n_local, n_total):
        # finished framebuffer -> rank 0: pack (tile order), weight, RCCL sum-reduce, unpack on rank 0
        ctx.pack_owned(red.data_ptr())
        ctx.sync()
        if args.dist_backend == "nccl":
            mg.reduce_samples(dist, red, n_local, n_total)
        else:
            host = red.cpu()
            mg.reduce_samples(dist, host, n_local, n_total)
            red.copy_(host)
        if rank == 0:
            torch.cuda.synchronize()
            ctx.unpack_from(red.data_ptr(), 0, 1)
            ctx.sync()

    ctx.render(0, mine[: args.warmup], args.bounces)
    if dist is not None:
        exchange(args.warmup, args.warmup * world)  # warms RCCL's lazy connection set-up too
        ctx.clear_framebuffer()
    sync_all()
    ctx.reset_counters()
    sync_all()
    t0 = time.perf_counter()
    # the local frame count sets the blend weight, so each rank's image is the mean of its own samples
    ctx.render(0 if dist is not None else args.warmup, mine[args.warmup:], args.bounces)
    if dist is not None:
        exchange(args.steps, args.steps * world)
    sync_all()
    dt = time.perf_counter() - t0
    cnt = ctx.counters()
    rays = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_bounce"]

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([rays, cnt["paths"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rays_total, paths_total = int(tot[0].item()), int(tot[1].item())
    else:
        rays_total, paths_total = rays, cnt["paths"]

    check = None
    if dist is not None and args.check and rank == 0:
        got = ctx.read_framebuffer()
        sctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, max_batch=args.max_batch)
        sctx.upload_scene(scene, tex_rgb8=tex)
        sctx.set_camera(cam)
        timed = np.concatenate([seeds[r::world][args.warmup:] for r in range(world)])  # the frames of the timed region
        sctx.render(0, timed, args.bounces)
        seq = sctx.read_framebuffer()
        sctx.close()
        err = np.abs(got[..., :3] - seq[..., :3]) / np.maximum(np.abs(seq[..., :3]), 1e-3)
        check = {"max_rel_err_vs_sequential": float(err.max()), "ok": bool(err.max() < 2e-6)}

    if rank != 0:
        ctx.close()
        dist.destroy_process_group()
        return

    # ---- roofline inputs for the dominant kernel (k_trace): S, T per ray from a counted replay of one batch
    cctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags | pt.PT_FLAG_COUNT_TRAVERSAL, max_batch=args.max_batch)
    cctx.upload_scene(scene, tex_rgb8=tex)
    cctx.set_camera(cam)
    nc = min(cctx.max_batch, args.steps)
    cctx.render(args.warmup, mine[args.warmup: args.warmup + nc], args.bounces)
    cc = cctx.counters()
    cctx.close()
    crays = cc["rays_primary"] + cc["rays_shadow"] + cc["rays_bounce"]
    S, T = cc["node_steps"] / crays, cc["tri_tests"] / crays
    bytes_per_ray = 64.0 * S + 64.0 * T + 96.0  # 64 B per node pair, 64 B per triangle, 96 B of queue traffic
    launches = max(cnt["trace_launches"], 1)
    avg_launch_ms = cnt["ms_trace"] / launches  # HIP events recorded around every k_trace launch on the context's stream
    rays_per_launch = rays / launches
    achieved = bytes_per_ray * rays_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    workload_key = f"{args.scene}-{args.width}x{args.height}-b{args.bounces}"
    pmc = pmc_traffic(workload_key)
    traffic = None
    if pmc is not None:
        k = pmc["kernels"]["k_trace"]
        # per ray as profiled, scaled to this run's rays per launch (batch sizes may differ between the two runs)
        traffic = round(k["hbm_bytes_per_launch"] / max(k["rays_per_launch"], 1.0) * rays_per_launch)
    roofline = {"bound": "hbm", "kernel": "k_trace", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": (pmc or {}).get("source"),
                "algorithmic_bytes_per_launch": round(bytes_per_ray * rays_per_launch),
                "bytes_per_ray": round(bytes_per_ray, 1), "node_steps_per_ray": round(S, 3), "tri_tests_per_ray": round(T, 3),
                "rays_per_launch": round(rays_per_launch, 1), "avg_launch_ms": round(avg_launch_ms, 4),
                "trace_ms_total": round(cnt["ms_trace"], 3), "shade_ms_total": round(cnt["ms_shade"], 3)}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle_lib as ol
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        osc = ol.scene_from_desc(scene.desc(), tex_rgb8=tex)
        nf = max(1, args.cpu_frames)
        c0 = time.perf_counter()
        _, oc = ol.render(osc, cam, args.width, args.height, pt.frame_seeds(args.warmup, nf), first_frame=args.warmup,
                          bounces=args.bounces, threads=threads)
        cdt = time.perf_counter() - c0
        orays = oc["rays_primary"] + oc["rays_shadow"] + oc["rays_bounce"]
        cpu = {"value": round(orays / cdt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": f"{nf} frame(s) of the same {args.width}x{args.height} workload (the oracle traces zero-throughput paths too)",
               "spp_per_s": round(nf / cdt, 4), "seconds": round(cdt, 2)}

    d = scene.desc()
    out = {
        "metric": f"Mrays/s (closest-hit + NEE shadow rays), MIS path, {'bunny-class' if args.scene.startswith('bunny') else args.scene} scene {args.width}x{args.height}",
        "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene}: {d.n_tris} triangles, Sweep-SAH BVH {d.n_nodes} nodes, {args.width}x{args.height}, "
                               f"{args.steps} spp per GPU, {args.bounces} bounces, MIS (BSDF + NEE), shadow rays closest-hit "
                               f"(reference semantics), {ctx.max_batch} frames per launch set",
                   "scene_hash": f"{scene.hash():016x}", "partition": "samples-interleaved" if world > 1 else "single",
                   "dead_paths": "traced" if args.trace_dead_paths else "retired"},
        "spp_per_s": round(args.steps * world / dt, 3), "mpaths_per_s": round(paths_total / dt / 1e6, 2),
        "rays": {"primary": cnt["rays_primary"], "shadow": cnt["rays_shadow"], "bounce": cnt["rays_bounce"], "scope": "rank0" if world > 1 else "all"},
        "stack_spills": cnt["stack_spills"], "roofline": roofline, "cpu_baseline": cpu,
        "setup": {"scene_generation_and_cpu_bvh_build_s": round(t_scene, 3), "upload_incl_traversal_blob_ms": round(t_upload * 1e3, 2)},
    }
    if check is not None:
        out["check"] = check
    print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
