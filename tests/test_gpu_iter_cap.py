"""Debug build of the library (`make -C pathtracer_amd/csrc debug`, -DPT_DEBUG_ITER_CAP): k_trace counts the iterations of
every ray and abandons one that passes 4 x (records of the traversal blob) + 64 -- a tree walk visits every record at most
once -- and the launch returns PT_EHIP with a message.  A traversal loop that does not end (commit b882dcd moved the step
calls behind an #else) is then a failing test, not a hung GPU.

Each case runs in a child process with PT355_LIB pointing at the debug library (one process loads one libpt355)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "pathtracer_amd", "libpt355_dbg.so")
STALL = os.path.join(ROOT, "pathtracer_amd", "libpt355_dbg_stall.so")

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import pathtracer_amd as pt, oracle_lib as ol
scene = pt.Scene.bunny(freq=10, target_tris=0, spheres=True)
cam, seeds = pt.default_camera(), pt.frame_seeds(0, 2)
c = pt.PathTracer(64, 64)
c.upload_scene(scene); c.set_camera(cam)
try:
    c.render(0, seeds, 5)
    img = c.read_framebuffer()
except pt.PtError as e:
    print("PTERROR:", e); sys.exit(3)
ref, _ = ol.render(ol.scene_from_desc(scene.desc()), cam, 64, 64, seeds)
print("IDENTICAL" if np.array_equal(img.view(np.uint32), ref.view(np.uint32)) else "DIFFERENT")
"""


def _child(lib):
    env = dict(os.environ, PT355_LIB=lib)
    return subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=120, env=env)


def test_cap_is_silent_on_a_healthy_traversal():
    if not os.path.exists(DBG):
        pytest.skip("debug library not built (make -C pathtracer_amd/csrc debug)")
    r = _child(DBG)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "IDENTICAL" in r.stdout


def test_a_loop_that_does_not_step_is_an_error_not_a_hang():
    if not os.path.exists(STALL):
        pytest.skip("debug library with the injected fault not built (make -C pathtracer_amd/csrc debug)")
    r = _child(STALL)   # every lane's first 100,000 iterations skip the step calls: far past any cap, yet finite
    assert r.returncode == 3, r.stdout[-2000:] + r.stderr[-2000:]
    assert "exceeded the traversal iteration cap" in r.stdout
