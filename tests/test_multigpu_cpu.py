"""The N > 1 path on CPU: world_size-2 and -3 gloo jobs run the partition / gather / reduce code that bench.py
runs over RCCL (pathtracer_amd.multigpu), with the oracle as the renderer."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_tiles_and_samples(tmp_path, world):
    out = tmp_path / "res.json"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res["world"] == world
    assert res["tiles_identical"] is True          # tiles: bit-identical to the 1-device image
    assert res["samples_alpha_one"] is True
    assert res["samples_max_rel_err"] < 2e-6       # samples: summation order differs from the running mean


def test_frames_of_rank_partition_the_range(pt):
    from pathtracer_amd import multigpu as mg
    for world in (1, 2, 3, 8):
        got = np.sort(np.concatenate([mg.frames_of_rank(5, 21, r, world) for r in range(world)]))
        assert got.tolist() == list(range(5, 26))


@pytest.mark.parametrize("w,h", [(8, 8), (40, 24), (1920, 1080)])
def test_tile_partition_covers_the_image_once(pt, w, h):
    for world in (1, 2, 3, 8):
        pix = [pt.partition_pixels(w, h, pt.PT_PARTITION_TILES, r, world) for r in range(world)]
        allp = np.concatenate(pix)
        assert allp.size == w * h and np.array_equal(np.sort(allp), np.arange(w * h, dtype=np.uint32))
        sizes = [p.size for p in pix]
        assert max(sizes) - min(sizes) <= 64  # balanced to one tile
        # tile t belongs to rank t % world, pixels inside a tile are row-major 8x8
        if (w // 8) * (h // 8) < world:
            continue
        t0 = pix[world - 1][:64]
        tile = world - 1
        x0, y0 = (tile % (w // 8)) * 8, (tile // (w // 8)) * 8
        want = [(y0 + i // 8) * w + x0 + i % 8 for i in range(64)]
        assert t0.tolist() == want
    full = pt.partition_pixels(w, h, pt.PT_PARTITION_FULL, 0, 1)
    assert np.array_equal(np.sort(full), np.arange(w * h, dtype=np.uint32))
    assert pt.lib().pt_partition_count(w + 1, h, 0, 0, 1) == 0 and pt.lib().pt_partition_count(w, h, 0, 2, 2) == 0
