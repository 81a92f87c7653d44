"""HIP path vs the committed golden vectors (tests/golden/oracle_*.npz): the same fixtures the CPU suite pins the
oracle with, so GPU == golden == oracle without the oracle having to run on the GPU box."""
import numpy as np
import pytest

from test_oracle import _scene, golden_env, golden_tex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cornell_spheres", "soup3000", "bunny12", "textured_spheres"])
def test_trace_golden(pt, golden, name):
    g = golden("oracle_trace.npz")
    sc = _scene(pt, name)
    assert sc.hash() == int(g[f"{name}_hash"][0])
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(sc)
        prim, th, kind, uv = c.trace_closest(g[f"{name}_ro"], g[f"{name}_rd"])
    finally:
        c.close()
    assert np.array_equal(kind, g[f"{name}_kind"])
    assert np.array_equal(prim, g[f"{name}_prim"])
    assert np.array_equal(th, g[f"{name}_th"])
    assert np.array_equal(uv.view(np.uint32), g[f"{name}_uv"].view(np.uint32))


@pytest.mark.parametrize("name", ["cornell", "cornell_spheres", "bunny12_on", "bunny12_ggx_env", "textured_spheres"])
def test_frame_golden(pt, golden, name):
    g = golden("oracle_frames.npz")
    sc = _scene(pt, name)
    assert sc.hash() == int(g[f"{name}_hash"][0])
    env = golden_env() if name.endswith("_env") else None
    c = pt.PathTracer(64, 64, flags=pt.PT_FLAG_TRACE_DEAD_PATHS | pt.PT_FLAG_COUNT_TRAVERSAL | pt.PT_FLAG_SHADOW_FULL_TRAVERSAL)
    try:
        c.upload_scene(sc, env_rgb8=env, tex_rgb8=golden_tex(pt, name))
        c.set_camera(pt.default_camera())
        c.render(0, pt.frame_seeds(0, 4), 5)
        img = c.read_framebuffer()
        cnt = c.counters()
    finally:
        c.close()
    assert np.array_equal(img.view(np.uint32), g[f"{name}_rgba"].view(np.uint32))
    # ray counts AND the traversal work counters S, T that define the algorithmic bytes
    assert [cnt[k] for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce", "node_steps", "tri_tests")] == g[f"{name}_counts"].tolist()
