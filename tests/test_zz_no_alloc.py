"""pt_trace_closest_device_noalloc (the entry point that allocates nothing; scratch in a file-scope device array)
against pt_trace_closest_device (scratch from hipMalloc) through the C ABI: the hit records must be identical bit for
bit -- prim u32, fp16 t | kind, (u, v) -- on two calls in a row, the second with a smaller input (the array keeps the
first call's contents), and the size limit must be an error, not a launch.

Collected last on purpose (test_zz_*): the round's end runs pytest -x."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays_soa(n, seed):
    rng = np.random.default_rng(seed)
    ro = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), rng.uniform(-7.9, -4.1, n)], 1).astype(np.float32)
    ro[: n // 2] = 0.0
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    return np.ascontiguousarray(np.concatenate([ro.T, rd.T], 0))  # [6, n]: ro.x ro.y ro.z rd.x rd.y rd.z


def test_noalloc_entry_matches_the_allocating_one(pt, ol):
    import torch

    scene = pt.Scene.bunny(spheres=True)
    c = pt.PathTracer(8, 8)
    try:
        c.upload_scene(scene)
        osc = ol.scene_from_desc(scene.desc())
        for n, seed in ((200000, 21), (4097, 22)):  # second call smaller: stale counters / spill of the first must not leak
            soa = _rays_soa(n, seed)
            d_ray = torch.from_numpy(soa).cuda()
            want = torch.full((n, 4), 0x7fffffff, dtype=torch.int32, device="cuda")
            got = torch.full((n, 4), 0x7ffffffe, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            c.trace_closest_device(d_ray.data_ptr(), n, want.data_ptr())
            c.sync()
            c.trace_closest_device_noalloc(d_ray.data_ptr(), n, got.data_ptr())
            c.sync()
            w, g = want.cpu().numpy().view(np.uint32), got.cpu().numpy().view(np.uint32)
            assert np.array_equal(g, w), f"{int((g != w).any(axis=1).sum())} of {n} hit records differ"
            # and both are the oracle's answer (bit-exact bar of the closest-hit contract)
            oprim, oth, okind, _ = ol.trace_closest(osc, np.ascontiguousarray(soa[:3].T), np.ascontiguousarray(soa[3:].T))
            kind = (g[:, 1] >> 16).astype(np.uint8)
            assert np.array_equal(kind, okind)
            hit = kind != 0
            assert np.array_equal(g[hit, 0], oprim[hit])
            assert np.array_equal((g[hit, 1] & 0xffff).astype(np.uint16), oth[hit])
            assert hit.mean() > 0.3
        # beyond the stated maximum: a clear error, nothing launched
        with pytest.raises(pt.PtError, match="maximum"):
            c.trace_closest_device_noalloc(d_ray.data_ptr(), (1 << 26) + 1, got.data_ptr())
    finally:
        c.close()


def test_noalloc_entry_never_synchronises_and_its_rate(pt):
    """With renders of the context in flight the entry point returns PT_ESTATE instead of waiting for them (a caller under
    graph capture cannot take a stream synchronisation); after pt_sync it runs.  Also prints its rate next to the
    allocating entry's: the grid is capped at 2,048 waves, a quarter of pt_create's (INTEGRATION.md quotes the figure)."""
    import time

    import torch

    scene = pt.Scene.bunny(spheres=True)
    c = pt.PathTracer(64, 64)
    try:
        c.upload_scene(scene)
        c.set_camera(pt.default_camera())
        n = 4 << 20
        d_ray = torch.from_numpy(_rays_soa(n, 5)).cuda()
        a = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        b = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        c.render(0, pt.frame_seeds(0, 2), 5)
        with pytest.raises(pt.PtError, match="pt_sync first"):
            c.trace_closest_device_noalloc(d_ray.data_ptr(), n, b.data_ptr())
        c.sync()
        rates = {}
        for name, fn, out in (("alloc", c.trace_closest_device, a), ("noalloc", c.trace_closest_device_noalloc, b)):
            fn(d_ray.data_ptr(), n, out.data_ptr())
            c.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                fn(d_ray.data_ptr(), n, out.data_ptr())
            c.sync()
            rates[name] = 5 * n / (time.perf_counter() - t0) / 1e9
        print(f"trace entries on {n} incoherent rays: pt_trace_closest_device {rates['alloc']:.2f} Grays/s, "
              f"pt_trace_closest_device_noalloc {rates['noalloc']:.2f} Grays/s ({rates['noalloc'] / rates['alloc']:.2f} x)")
        assert torch.equal(a, b)
    finally:
        c.close()
