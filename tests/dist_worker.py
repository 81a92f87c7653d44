"""Worker for tests/test_multigpu_cpu.py: one rank of a world_size-N gloo job on CPU.  Exercises the partition and
exchange code of pathtracer_amd.multigpu with the CPU oracle standing in for the HIP renderer (test infrastructure)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PT355_NO_TORCH"] = "0"

import oracle_lib as ol  # noqa: E402
import pathtracer_amd as pt  # noqa: E402
from pathtracer_amd import multigpu as mg  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H, NF = 40, 24, 6
    sc = pt.Scene.cornell(spheres=True)
    osc, cam = ol.scene_from_desc(sc.desc()), pt.default_camera()
    seeds = pt.frame_seeds(0, NF)
    full, _ = ol.render(osc, cam, W, H, seeds)  # the sequential 1-device answer

    # ---- tiles: every rank renders all frames, keeps its own pixels, rank 0 gathers
    mine = pt.partition_pixels(W, H, pt.PT_PARTITION_TILES, rank, world)
    n_max = max(pt.partition_pixels(W, H, pt.PT_PARTITION_TILES, r, world).size for r in range(world))
    packed = torch.zeros((n_max, 4), dtype=torch.float32)
    packed[: mine.size] = torch.from_numpy(full.reshape(-1, 4)[mine])
    bufs = mg.gather_tiles(dist, packed, rank, world)
    res = {}
    if rank == 0:
        img = mg.assemble_tiles(bufs, W, H, world, pt.partition_pixels)
        res["tiles_identical"] = bool(np.array_equal(img.view(np.uint32), full.view(np.uint32)))

    # ---- samples: frames dealt round-robin, local running means, one sum-reduce
    acc = np.zeros((H, W, 4), np.float32)
    k = 0
    for f in mg.frames_of_rank(0, NF, rank, world):
        acc, _ = ol.render(osc, cam, W, H, seeds[f: f + 1], first_frame=k, rgba=acc)
        k += 1
    t = torch.from_numpy(acc.copy())
    mg.reduce_samples(dist, t, k, NF)
    if rank == 0:
        got = t.numpy()
        err = np.abs(got[..., :3] - full[..., :3]) / np.maximum(np.abs(full[..., :3]), 1e-3)
        res["samples_max_rel_err"] = float(err.max())
        res["samples_alpha_one"] = bool((got[..., 3] == 1.0).all())
        res["world"] = world
        import json
        with open(out_path, "w") as fh:
            json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
