#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (closest-hit + NEE shadow rays) of the MIS path on the bunny-class scene at
1920x1080 (BASELINE.json configs[1]).  One step = one frame = one sample per pixel over the whole image.

    python bench.py --gpus N --steps K --warmup W [--partition samples|tiles] [--scene NAME|file.glb]

N > 1: one rank per GPU.  Started from a plain shell (no WORLD_SIZE in the environment) the script starts its own
N ranks -- a child `python -m torch.distributed.run ... bench.py <same arguments>`, spawned BEFORE anything in this
process touches the GPU -- and exits with the child's code; started by torch.distributed.run it is one rank.
The path shards with no data-path collective; the scene is replicated; the only exchange is the finished framebuffer,
inside the timed region.  Two partitions:

  samples (default; weak scaling, configs[4])   global frame f -> rank f % N over the FULL image; every rank renders K
            frames with the host-PCG seeds of its own global frame numbers, N*K samples per pixel in all; exchange =
            one RCCL sum-reduce of (n_r / N_total) * mean_r to rank 0.
  tiles   (strong scaling, configs[3])           8x8 tiles dealt round-robin, every rank renders all K frames for its own
            pixels; the image is bit-identical to the 1-GPU image; exchange = one gather of packed tiles to rank 0.

Timing: W warm-up frames (plus one untimed batch of the timed launch size when W is smaller), then `--repeats`
(default 3) timed regions of EXACTLY K steps each, bracketed by barrier + synchronize, MAX over ranks; the MEDIAN
region is reported.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# k_trace's cache-miss read requests per global record fetch on the bunny workload (profiles/r02d_pmc_traffic.json, k_trace_all);
# used for the gather ceiling only when the workload being benched has no PMC passes of its own
DEFAULT_MISS_L1, DEFAULT_MISS_L2 = 0.57, 0.11
BVH_NAMES = {"reference": "Sweep-SAH", "lbvh": "GPU-built LBVH", "ploc": "GPU-built PLOC"}


def make_scene(pt, name):
    if name == "bunny":      # configs[1]: 69,451-triangle stand-in mesh, Oren-Nayar, in the Cornell shell
        return pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_OREN_NAYAR, roughness=0.175)
    if name == "bunny_ggx":  # configs[2]
        return pt.Scene.bunny(freq=59, target_tris=69451, seed=1, material=pt.MAT_SPECULAR_METAL, roughness=0.2, spheres=True)
    if name == "atrium":     # configs[3] stand-in: 281,620 triangles, 28 materials, texture layers 0..5
        return pt.Scene.atrium(detail=4, seed=1)
    if name == "cornell":    # configs[0]
        return pt.Scene.cornell()
    if name == "atrium_gltf":  # configs[3] through the glTF path: the atrium written out as a .glb, then LoadGLTF + src/main.cpp's set-up
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_scene_glb
        path = os.path.join(tempfile.gettempdir(), f"pt355_atrium_{os.getpid()}.glb")
        make_scene_glb.write_scene_glb(pt.Scene.atrium(detail=4, seed=1), path, pt.make_textures(6, 1))
        sc = pt.Scene.from_gltf(path)
        os.unlink(path)
        return sc
    if name.endswith((".glb", ".gltf")) and hasattr(pt.Scene, "from_gltf"):
        return pt.Scene.from_gltf(name)  # the reference's LoadGLTF + the scene set-up of src/main.cpp:19-63
    raise SystemExit(f"unknown scene {name}")


def pmc_profile(workload_key, kernel_sha):
    """Per-launch PMC figures of k_trace from the committed rocprofv3 --pmc passes of this same workload
    (profiles/*_pmc_traffic.json, written by tools/prof_summary.py): bench.py cannot run the profiler on itself.
    Only a profile stamped with the SHA-256 of the device sources this run executes counts (`kernel_sha256`, written by
    tools/prof_summary.py): returns (profile or None, reason when None)."""
    import glob
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload_key and "k_trace" in d.get("kernels", {}):
            if d.get("kernel_sha256") == kernel_sha:
                return d, None
            stale = stale or os.path.basename(f)
    if stale:
        return None, (f"no PMC profile of the kernel sources this run executes (sha256 {kernel_sha[:12]}...): the newest one for this workload, "
                      f"profiles/{stale}, was taken on other sources -- re-run tools/prof.sh and commit its summary")
    return None, f"no PMC profile for workload {workload_key} under profiles/"


def gather_ceiling(device, miss_l1, miss_l2, active_lanes, waves_per_simd):
    """Live measurement on this GPU (pathtracer_amd/libptubench.so = tools/ubench_gather.hip): lane-steps/s of dependent
    64-byte record gathers at the trace kernel's cache-miss rates per record fetch (TCP->TCC and TCC->fabric read requests
    per lane-step, from the PMC passes), its lane occupancy and resident waves, and no arithmetic in between.
    Returns (lane_steps_per_s, ns_per_wavestep) or None when the library is not built."""
    path = os.path.join(ROOT, "pathtracer_amd", "libptubench.so")
    if not os.path.exists(path):
        return None
    L = ctypes.CDLL(path)
    L.ptu_gather_ceiling.restype = ctypes.c_int
    L.ptu_gather_ceiling.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    ns, rate = ctypes.c_double(), ctypes.c_double()
    rc = L.ptu_gather_ceiling(device, float(miss_l1), float(miss_l2), int(active_lanes), int(waves_per_simd), 2000,
                              ctypes.byref(ns), ctypes.byref(rate))
    return (rate.value, ns.value) if rc == 0 else None


def spawn_ranks(n):
    """Plain-shell start with --gpus N > 1: run the N ranks as a child torch.distributed.run job.  Nothing in this
    process has touched the GPU yet (a process that has must never be replaced or forked into GPU work)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps frames each; the median is reported")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--bvh", choices=["reference", "lbvh", "ploc"], default="reference",
                    help="reference = the reference's own Sweep-SAH builder (what its renders use; the default and the parity contract); "
                         "lbvh / ploc = the GPU builders of pt_build_bvh_device (max_leaf 2)")
    ap.add_argument("--partition", choices=["samples", "tiles"], default="samples",
                    help="N > 1: samples = frames dealt round-robin (weak scaling), tiles = 8x8 screen tiles (strong scaling, bit-identical)")
    ap.add_argument("--max-batch", type=int, default=0, help="frames traced per launch set (0 = library default)")
    ap.add_argument("--trace-dead-paths", action="store_true", help="trace zero-throughput paths like the reference does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=48)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; what the driver runs) or gloo (rehearsal of the N > 1 "
                    "code path with several ranks sharing one GPU: the exchange then goes through host memory)")
    ap.add_argument("--check", action="store_true", help="kept for old command lines: at N > 1 the check now always runs (see --no-check)")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the comparison of the exchanged image with a sequential render of the "
                    "same frames on rank 0 (tiles: bit-identical; samples: 2e-6 relative); it runs once, outside the timed regions")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of back-to-back K-step regions after the timed ones (0 = off): "
                    "`sustained_ms_per_step` is long enough to be confirmed by an outside clock")
    ap.add_argument("--no-anyhit", action="store_true", help="skip the PT_FLAG_SHADOW_ANYHIT leg (rate + image RMSE against the default mode)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if os.environ.get("PT355_BENCH_WATCHDOG"):
        # a rank that hangs (a collective one side never enters) prints every thread's stack and exits instead of sitting silent
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PT355_BENCH_WATCHDOG"]), exit=True)

    import numpy as np
    import torch
    import pathtracer_amd as pt
    from pathtracer_amd import multigpu as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py (rank {rank} of {world}) needs a HIP device: the path has no CPU fallback")
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
    tiles = world > 1 and args.partition == "tiles"
    # where the few-byte control reductions (times, counts) live: RCCL reduces device tensors; gloo is given host tensors -- its
    # CUDA path deadlocked now and then with two ranks on one GPU (both ranks inside the same all_reduce, r03 watchdog dump)
    ctl_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    K, W = args.steps, args.warmup

    t_scene = time.perf_counter()
    scene = make_scene(pt, args.scene)  # generator + the reference's single-threaded Sweep-SAH build (CalculateBVH)
    t_scene = time.perf_counter() - t_scene
    flags = pt.PT_FLAG_TRACE_DEAD_PATHS if args.trace_dead_paths else 0
    t_gpu_bvh_ms = None
    if args.bvh != "reference":
        bctx = pt.PathTracer(8, 8, device=local_rank)
        t_gpu_bvh_ms = bctx.finalize_scene_device(scene, 2, ploc=args.bvh == "ploc")
        bctx.close()
    if tiles:
        ctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, partition=pt.PT_PARTITION_TILES, rank=rank, world=world,
                            max_batch=args.max_batch)
    else:
        ctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, partition=pt.PT_PARTITION_FULL, max_batch=args.max_batch)
    tex = pt.make_textures(6, 1) if args.scene == "atrium" else getattr(scene, "textures", None)
    tex_sampler = getattr(scene, "tex_sampler", None)
    t_upload = time.perf_counter()
    ctx.upload_scene(scene, tex_rgb8=tex, tex_sampler=tex_sampler)
    ctx.sync()
    t_upload = time.perf_counter() - t_upload
    cam = pt.default_camera()
    ctx.set_camera(cam)
    # one seed table for the whole job, indexed by GLOBAL frame number
    if tiles or world == 1:
        seeds = pt.frame_seeds(0, W + K)
        mine = seeds                       # every rank renders every frame (for its own pixels)
    else:
        seeds = pt.frame_seeds(0, (W + K) * world)
        mine = seeds[rank::world]          # rank r renders global frames r, r + world, ...
    n_pix = args.width * args.height
    red = None
    if dist is not None:
        rows = n_pix if not tiles else max(pt.lib().pt_partition_count(args.width, args.height, pt.PT_PARTITION_TILES, r, world) for r in range(world))
        red = torch.zeros((rows, 4), dtype=torch.float32, device="cuda")
    full_ctx = None  # tiles: rank 0 assembles the gathered image in a full-image context of its own

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            if args.dist_backend == "nccl":
                dist.barrier()
            else:
                dist.all_reduce(torch.zeros(1))  # a host-side barrier: gloo's barrier() picks the CUDA path when a device is current
            torch.cuda.synchronize()

    def exchange(n_local, n_total):
        # finished framebuffer -> rank 0
        ctx.pack_owned(red.data_ptr())
        ctx.sync()
        if tiles:
            # gather of packed tiles (rows past a rank's own count are padding), then unpack rank by rank
            if args.dist_backend == "nccl":
                bufs = mg.gather_tiles(dist, red, rank, world)
            else:
                hb = mg.gather_tiles(dist, red.cpu(), rank, world)
                bufs = [b.cuda() for b in hb] if hb is not None else None
            if rank == 0:
                torch.cuda.synchronize()
                for r in range(world):
                    full_ctx.unpack_from(bufs[r].data_ptr(), r, world)
                full_ctx.sync()
            return
        # samples: pack (tile order), weight, RCCL sum-reduce, unpack on rank 0
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

    if tiles and rank == 0:
        full_ctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, partition=pt.PT_PARTITION_FULL, max_batch=1)

    # ---- warm-up: W frames, and at least one launch set of the timed size (a smaller warm-up launches smaller kernels)
    timed = mine[W: W + K]
    ctx.render(0, mine[:W], args.bounces)
    nb = min(ctx.max_batch, K)
    if W < nb:
        ctx.render(W, timed[:nb], args.bounces)
    if dist is not None:
        exchange(max(W, 1), max(W, 1) * (1 if tiles else world))  # warms RCCL's lazy connection set-up too
    sync_all()

    regions = []
    for rep in range(max(1, args.repeats)):
        ctx.clear_framebuffer()
        ctx.reset_counters()
        sync_all()
        t0 = time.perf_counter()
        # the local frame count sets the blend weight, so each rank's image is the mean of its own samples
        ctx.render(0, timed, args.bounces)
        if dist is not None:
            exchange(K, K * (1 if tiles else world))
        sync_all()
        dt = time.perf_counter() - t0
        cnt = ctx.counters()
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        regions.append((dt, cnt))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, cnt = regions[order[len(order) // 2]]
    rays = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_bounce"]

    # ---- sustained leg: the same K-step region back to back until at least --sustain seconds have passed, so that the time
    # per step can be confirmed from outside the process (the driver's wall clock, GPU-busy samples): 3 x K steps are a few
    # tens of milliseconds.  Whole regions only; timed like the regions above; reported beside the median, never instead of it.
    sustained = None
    if args.sustain > 0:
        sync_all()
        s0 = time.perf_counter()
        n_reg = 0
        while True:
            ctx.clear_framebuffer()  # as in the timed regions (frame 0 overwrites the image anyway, except where it holds NaN)
            ctx.render(0, timed, args.bounces)
            if dist is not None:
                exchange(K, K * (1 if tiles else world))
            n_reg += 1
            ctx.sync()
            el = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device=ctl_dev)
            if dist is not None:
                dist.all_reduce(el, op=dist.ReduceOp.MAX)  # every rank takes the same decision
            if float(el.item()) >= args.sustain or n_reg >= 100000:
                break
        sync_all()
        sdt = time.perf_counter() - s0
        if dist is not None:
            tmax = torch.tensor([sdt], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            sdt = float(tmax.item())
        sustained = {"seconds": round(sdt, 3), "regions": n_reg, "steps": n_reg * K, "ms_per_step": round(sdt / (n_reg * K) * 1e3, 4)}

    if dist is not None:
        tot = torch.tensor([rays, cnt["paths"]], dtype=torch.int64, device=ctl_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rays_total, paths_total = int(tot[0].item()), int(tot[1].item())
    else:
        rays_total, paths_total = rays, cnt["paths"]

    # ---- N > 1: who took part, and did N ranks on N devices produce the right image?  Outside the timed regions.
    ranks_seen = None
    if dist is not None:
        props = torch.cuda.get_device_properties(local_rank)
        me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "device": torch.cuda.get_device_name(local_rank),
              "pci_bus_id": "%04x:%02x:%02x" % (int(getattr(props, "pci_domain_id", 0)), int(getattr(props, "pci_bus_id", 0)), int(getattr(props, "pci_device_id", 0))),
              "uuid": str(getattr(props, "uuid", "")), "backend": args.dist_backend}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
    check = None
    if dist is not None and not args.no_check and rank == 0:
        got = (full_ctx if tiles else ctx).read_framebuffer()
        sctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags, max_batch=args.max_batch)
        sctx.upload_scene(scene, tex_rgb8=tex, tex_sampler=tex_sampler)
        sctx.set_camera(cam)
        if tiles:
            frames = timed
        else:
            frames = np.concatenate([seeds[r::world][W: W + K] for r in range(world)])  # the frames of the timed region
        sctx.render(0, frames, args.bounces)
        seq = sctx.read_framebuffer()
        sctx.close()
        if tiles:
            bad = int((got.view(np.uint32) != seq.view(np.uint32)).any(axis=2).sum())
            check = {"differing_pixels_vs_1gpu": bad, "ok": bad == 0}
        else:
            err = np.abs(got[..., :3] - seq[..., :3]) / np.maximum(np.abs(seq[..., :3]), 1e-3)
            check = {"max_rel_err_vs_sequential": float(err.max()), "ok": bool(err.max() < 2e-6)}

    if rank != 0:
        ctx.close()
        dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (k_trace): S, T per ray and lane occupancy from a counted replay of one batch
    cctx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags | pt.PT_FLAG_COUNT_TRAVERSAL, partition=ctx.partition,
                         rank=ctx.rank, world=ctx.world, max_batch=args.max_batch)
    cctx.upload_scene(scene, tex_rgb8=tex, tex_sampler=tex_sampler)
    cctx.set_camera(cam)
    cctx.render(0, timed[: min(cctx.max_batch, K)], args.bounces)
    cc = cctx.counters()
    trace_grid = int(cctx.info[5])
    cctx_info = [int(v) for v in cctx.info]
    cctx.close()
    crays = cc["rays_primary"] + cc["rays_shadow"] + cc["rays_bounce"]
    S, T = cc["node_steps"] / crays, cc["tri_tests"] / crays
    lane_occ = (cc["node_steps"] + cc["tri_tests"]) / max(64.0 * cc["wave_iters"], 1.0)  # share of lanes working per traversal iteration
    top_share = cc.get("top_steps", 0) / max(cc["node_steps"] + cc["tri_tests"], 1)   # lane-steps served by the LDS copy of the tree top
    launches = max(cnt["trace_launches"], 1)
    avg_launch_ms = cnt["ms_trace"] / launches  # HIP events recorded around every k_trace launch on the context's stream
    rays_per_launch = rays / launches
    sec = avg_launch_ms * 1e-3
    workload_key = f"{args.scene}-{args.width}x{args.height}-b{args.bounces}" + ("" if args.bvh == "reference" else f"-{args.bvh}")
    kernel_sha = pt.kernel_source_sha256()
    pmc, pmc_missing = pmc_profile(workload_key, kernel_sha)
    kern = (pmc or {}).get("kernels", {})
    pk = kern.get("k_trace_all") or kern.get("k_trace", {})  # all trace launches of a batch, like the HIP-event average here
    traffic = None
    if pk.get("hbm_bytes_per_launch"):
        # per ray as profiled, scaled to this run's rays per launch (batch sizes may differ between the two runs)
        traffic = round(pk["hbm_bytes_per_launch"] / max(pk.get("rays_per_launch", 0.0), 1.0) * rays_per_launch)
    # (1) the two resources the kernel leans on, each against its own peak; `roofline` reports the busier one.
    #  * VALU issue: wave64 vector instructions per second (SQ_INSTS_VALU of the committed PMC passes of this workload,
    #    scaled per ray to this run) against 1 instruction per 2 cycles per SIMD (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cyc, SIMD-32);
    #  * vector-memory gathers: lane-steps that fetch their 64-B record from global memory (the others read the LDS copy of
    #    the tree top) per second, against the live micro-benchmark at the kernel's own cache-miss rates and lane count.
    d = scene.desc()
    props = torch.cuda.get_device_properties(local_rank)
    n_cu = props.multi_processor_count
    trace_block = int(cctx_info[6]) or 256
    waves_per_simd = max(1, min(8, round(trace_grid * (trace_block // 64) / max(4 * n_cu, 1))))
    have_pmc = "record_miss_l1_per_lane_step" in pk
    miss_l1 = pk.get("record_miss_l1_per_lane_step", DEFAULT_MISS_L1)
    miss_l2 = pk.get("record_miss_l2_per_lane_step", DEFAULT_MISS_L2)
    lane_steps = (S + T) * rays_per_launch
    global_steps = lane_steps * (1.0 - top_share)
    achieved_ls = global_steps / sec / 1e9 if sec > 0 else 0.0
    gather_lanes = max(1, round(lane_occ * 64 * (1.0 - top_share)))
    ceil = gather_ceiling(local_rank, miss_l1, miss_l2, gather_lanes, waves_per_simd)
    pmc_name = ("PMC " + pmc["source"].split(":")[0]) if pmc else None
    vmem = {"bound": "vmem-gather", "achieved": round(achieved_ls, 2), "peak": round(ceil[0] / 1e9, 2) if ceil else None, "unit": "G lane-steps/s",
            "frac": round(achieved_ls / (ceil[0] / 1e9), 4) if ceil else None,
            "peak_source": ("measured live by pathtracer_amd/libptubench.so (tools/ubench_gather.hip: ptu_gather_ceiling): dependent 64-B "
                            f"record gathers, 4 x dwordx4 per lane-step, at the kernel's cache-miss requests per record fetch -- {miss_l1:.3f} "
                            f"TCP->TCC and {miss_l2:.3f} TCC->fabric read requests per global lane-step ("
                            f"{pmc_name if have_pmc else 'no PMC passes for this workload: the bunny workload figures of profiles/r02d_pmc_traffic.json'}), "
                            f"{gather_lanes} of 64 lanes gathering per wave-step (counted replay), {waves_per_simd} waves/SIMD, no arithmetic"
                            if ceil else "libptubench.so not built"),
            "record_miss_l1_per_lane_step": round(miss_l1, 4), "record_miss_l2_per_lane_step": round(miss_l2, 4),
            "peak_ns_per_wavestep": round(ceil[1], 2) if ceil else None,
            "lane_steps_from_lds_share": round(top_share, 4)}
    clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
    valu_peak = 4.0 * n_cu * clock_hz / 2.0 / 1e9  # G wave-instructions/s
    valu = None
    if pk.get("sq_insts_valu") and pk.get("rays_per_launch"):
        insts = pk["sq_insts_valu"] / pk["rays_per_launch"] * rays_per_launch  # same code, same scene: instructions scale with the rays
        ach = insts / sec / 1e9 if sec > 0 else 0.0
        valu = {"bound": "valu-issue", "achieved": round(ach, 2), "peak": round(valu_peak, 2), "unit": "G wave-instr/s", "frac": round(ach / valu_peak, 4),
                "peak_source": f"4 SIMDs x {n_cu} CUs x {clock_hz / 1e9:.2f} GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md, SIMD-32); "
                               f"instruction count: SQ_INSTS_VALU per ray of {pmc_name}",
                "valu_insts_per_launch": round(insts),
                "valu_lane_utilisation": round(pk["sq_thread_cycles_valu"] / (64.0 * pk["sq_active_inst_valu"]), 4) if pk.get("sq_active_inst_valu") else None}
        if valu["valu_lane_utilisation"] is not None:
            # issue slots that did work for a ray: the kernel issues `frac` of the peak's wave-instructions, and of the 64 lanes of
            # an issued instruction only this share is switched on (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU)
            valu["valu_useful_frac"] = round(valu["frac"] * valu["valu_lane_utilisation"], 4)
            valu["peak_source"] += "; valu_useful_frac = frac x valu_lane_utilisation = the share of the peak's LANE-slots that carry a ray's arithmetic"
    # (2) SURVEY.md 8(d)'s figure, kept for the record: algorithmic bytes against the HBM peak.  The 7 MB blob is
    # cache-served, so this "fraction" exceeds 1 and is NOT the binding roof; hbm_counter_frac is the HBM-side truth.
    bytes_per_ray = 64.0 * S + 64.0 * T + 96.0
    alg = bytes_per_ray * rays_per_launch / sec / 1e9 if sec > 0 else 0.0
    # the two fractions sit within a point or two of each other (0.64 / 0.65): inside a band of 0.02 the line leads with VALU issue, the
    # resource the experiments single out (EXPERIMENTS.md section 1), so that the headline does not flip between runs
    lead = valu if (valu and (vmem["frac"] is None or valu["frac"] >= vmem["frac"] - 0.02)) else vmem
    other = vmem if lead is valu else valu
    roofline = {"kernel": "k_trace"}
    roofline.update(lead)
    roofline.update({
                "second_resource": other,
                "lane_occupancy": round(lane_occ, 4), "lane_steps_per_ray": round(S + T, 3),
                "traffic": traffic, "traffic_source": (pmc or {}).get("source"),
                "hbm_counter_frac": round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic and sec > 0 else None,
                "algorithmic_hbm": {"bound": "hbm", "achieved": round(alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / HBM_PEAK_GBS, 5),
                                    "note": "SURVEY.md 8(d): (64 S + 64 T + 96) B per ray over the kernel time; above 1 because the scene is cache-resident",
                                    "algorithmic_bytes_per_launch": round(bytes_per_ray * rays_per_launch), "bytes_per_ray": round(bytes_per_ray, 1)},
                "node_steps_per_ray": round(S, 3), "tri_tests_per_ray": round(T, 3),
                "rays_per_launch": round(rays_per_launch, 1), "avg_launch_ms": round(avg_launch_ms, 4),
                "trace_ms_total": round(cnt["ms_trace"], 3), "shade_ms_total": round(cnt["ms_shade"], 3)})
    roofline["kernel_sha256"] = kernel_sha
    if pmc is None:
        # the instruction counts and cache-miss rates that price the roofline come from PMC passes; without passes of THESE sources
        # there is no fraction to report (the live figures below -- S, T, launch time, algorithmic bytes -- stand)
        roofline.update({"bound": "valu-issue", "achieved": None, "peak": round(valu_peak, 2), "unit": "G wave-instr/s", "frac": None,
                         "frac_withheld": pmc_missing, "second_resource": vmem})
    # the other third of the frame: k_shade (+ its primary form and k_resolve) IS bound by HBM bytes -- PMC bytes per launch of the same
    # workload over the HIP-event time of these kernels in the timed region
    if have_pmc and all(k in pmc["kernels"] for k in ("k_shade", "k_shade<primary>", "k_resolve")) and cnt["ms_shade"] > 0:
        n_sets = launches / float(args.bounces + 1)  # launch sets (batches) in the timed region
        rscale = rays_per_launch / max(pk.get("rays_per_launch", 0.0), 1.0)  # as for `traffic`: the profiled batch may differ in size
        sbytes = rscale * n_sets * (args.bounces * pmc["kernels"]["k_shade"]["hbm_bytes_per_launch"] + pmc["kernels"]["k_shade<primary>"]["hbm_bytes_per_launch"]
                           + pmc["kernels"]["k_resolve"]["hbm_bytes_per_launch"])
        sgbs = sbytes / (cnt["ms_shade"] * 1e-3) / 1e9
        roofline["shading_kernels"] = {"kernels": "k_shade<primary> + bounces x k_shade + k_resolve", "bound": "hbm", "achieved": round(sgbs, 1), "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": round(sgbs / HBM_PEAK_GBS, 4), "traffic_per_launch_set": round(sbytes / n_sets),
                                       "note": "HBM-side bytes from the PMC passes (same source as `traffic`); a plain 32-column read+write copy reaches 5350 GB/s on this chip (tools/ubench_soa.hip)"}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle_lib as ol
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        osc = ol.scene_from_desc(scene.desc(), tex_rgb8=tex)
        nf = max(1, args.cpu_frames)
        c0 = time.perf_counter()
        _, oc = ol.render(osc, cam, args.width, args.height, pt.frame_seeds(W, nf), first_frame=W,
                          bounces=args.bounces, threads=threads)
        cdt = time.perf_counter() - c0
        orays = oc["rays_primary"] + oc["rays_shadow"] + oc["rays_bounce"]
        cpu = {"value": round(orays / cdt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": f"{nf} frame(s) of the same {args.width}x{args.height} workload (the oracle traces zero-throughput paths too)",
               "spp_per_s": round(nf / cdt, 4), "seconds": round(cdt, 2)}

    # ---- the fast shadow mode (hard part 2): the headline number is the reference's closest-hit shadow semantics
    # (fc:970-974); PT_FLAG_SHADOW_ANYHIT stops a shadow ray at ANY hit in front of the light sample.  Its rate, and how far its
    # image is from the default mode's on the SAME frames, stated beside the headline -- never as the headline.
    anyhit = None
    if world == 1 and not args.no_anyhit:
        base_img = ctx.read_framebuffer()  # the K timed frames in the default mode (every region renders the same frames)
        actx = pt.PathTracer(args.width, args.height, device=local_rank, flags=flags | pt.PT_FLAG_SHADOW_ANYHIT, max_batch=args.max_batch)
        actx.upload_scene(scene, tex_rgb8=tex, tex_sampler=tex_sampler)
        actx.set_camera(cam)
        actx.render(0, timed[: min(actx.max_batch, K)], args.bounces)  # warm-up: one launch set of the timed size
        adts = []
        for _ in range(max(1, args.repeats)):
            actx.clear_framebuffer()
            actx.reset_counters()
            actx.sync()
            a0 = time.perf_counter()
            actx.render(0, timed, args.bounces)
            actx.sync()
            adts.append(time.perf_counter() - a0)
        acnt = actx.counters()
        any_img = actx.read_framebuffer()
        actx.close()
        adt = sorted(adts)[len(adts) // 2]
        arays = acnt["rays_primary"] + acnt["rays_shadow"] + acnt["rays_bounce"]
        mean = float(base_img[..., :3].mean())
        rmse = float(np.sqrt(np.mean((any_img[..., :3].astype(np.float64) - base_img[..., :3].astype(np.float64)) ** 2)))
        anyhit = {"flag": "PT_FLAG_SHADOW_ANYHIT", "value": round(arays / adt / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(adt / K * 1e3, 4),
                  "vs_closest_hit_mode": round((arays / adt) / (rays_total / dt), 4),
                  "image_rmse_vs_default": rmse, "image_rmse_rel_to_mean_radiance": round(rmse / mean, 6) if mean > 0 else None,
                  "mean_radiance_default": mean, "mean_radiance_anyhit": float(any_img[..., :3].mean()),
                  "differing_pixels": int((any_img.view(np.uint32) != base_img.view(np.uint32)).any(axis=2).sum()),
                  "frames_compared": K, "rays": arays,
                  "note": "same frames and seeds in both modes; any-hit ignores the one-sided compare of fc:974, so it can only remove light"}

    spp_total = K * (1 if (tiles or world == 1) else world)
    out = {
        "metric": f"Mrays/s (closest-hit + NEE shadow rays), MIS path, {'bunny-class' if args.scene.startswith('bunny') else os.path.basename(args.scene)} scene {args.width}x{args.height}",
        "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene}: {d.n_tris} triangles, {BVH_NAMES[args.bvh]} BVH {d.n_nodes} nodes, {args.width}x{args.height}, "
                               f"{K} spp per GPU, {args.bounces} bounces, MIS (BSDF + NEE), shadow rays closest-hit "
                               f"(reference semantics), {ctx.max_batch} frames per launch set",
                   "scene_hash": f"{scene.hash():016x}",
                   "partition": ("tiles" if tiles else "samples-interleaved") if world > 1 else "single",
                   "dead_paths": "traced" if args.trace_dead_paths else "retired"},
        "repeats": len(regions), "region_ms": [round(r[0] * 1e3, 3) for r in regions],
        "spp_per_s": round(spp_total / dt, 3), "mpaths_per_s": round(paths_total / dt / 1e6, 2),
        "rays": {"primary": cnt["rays_primary"], "shadow": cnt["rays_shadow"], "bounce": cnt["rays_bounce"], "scope": "rank0" if world > 1 else "all"},
        "stack_spills": cnt["stack_spills"], "roofline": roofline, "cpu_baseline": cpu,
        "setup": {"scene_generation_and_cpu_bvh_build_s": round(t_scene, 3), "upload_incl_traversal_blob_ms": round(t_upload * 1e3, 2),
                  "gpu_bvh_build_ms": round(t_gpu_bvh_ms, 3) if t_gpu_bvh_ms is not None else None},
    }
    out["sustained_ms_per_step"] = sustained["ms_per_step"] if sustained else None
    out["sustained"] = sustained
    out["world_size"] = world
    if ranks_seen is not None:
        out["ranks_seen"] = ranks_seen
        out["distinct_devices"] = len({(r["host"], r["pci_bus_id"], r["uuid"]) for r in ranks_seen})
    if check is not None:
        out["check"] = check
    if anyhit is not None:
        out["anyhit"] = anyhit
    print(json.dumps(out))
    sys.stdout.flush()
    ctx.close()
    if full_ctx is not None:
        full_ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
