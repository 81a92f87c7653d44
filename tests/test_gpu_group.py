"""pt_group (native multi-device behind the C ABI) on ONE GPU with N virtual devices: the same
partition / pack / peer-copy / unpack / weighted-sum code that runs over N real devices, minus the xGMI hop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, NF = 256, 128, 7


def _single(pt, scene, seeds):
    c = pt.PathTracer(W, H, max_batch=4)
    try:
        c.upload_scene(scene)
        c.set_camera(pt.default_camera())
        c.render(0, seeds, 5)
        return c.read_framebuffer(), c.counters()
    finally:
        c.close()


@pytest.mark.parametrize("n", [2, 3])
def test_tiles_group_is_bit_identical_to_one_device(pt, n):
    scene = pt.Scene.bunny(freq=12, target_tris=0, spheres=True)
    seeds = pt.frame_seeds(0, NF)
    ref, rc = _single(pt, scene, seeds)
    g = pt.Group(W, H, [0] * n, mode=pt.PT_GROUP_TILES, max_batch=4)
    try:
        g.upload_scene(scene)
        g.set_camera(pt.default_camera())
        g.render(0, seeds[:3], 5)          # two calls: the running mean continues across them
        g.render(3, seeds[3:], 5)
        img = g.read_framebuffer()
        gc = g.counters()
    finally:
        g.close()
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    for k in ("paths", "rays_primary", "rays_shadow", "rays_bounce"):
        assert gc[k] == rc[k]                # the members' work adds up to the single device's, ray for ray
    assert gc["frames"] == NF


@pytest.mark.parametrize("n", [2, 3])
def test_samples_group_matches_the_sequential_mean(pt, n):
    scene = pt.Scene.cornell(spheres=True)
    seeds = pt.frame_seeds(0, NF)
    ref, rc = _single(pt, scene, seeds)
    g = pt.Group(W, H, [0] * n, mode=pt.PT_GROUP_SAMPLES, max_batch=4)
    try:
        g.upload_scene(scene)
        g.set_camera(pt.default_camera())
        g.render(0, seeds, 5)
        img = g.read_framebuffer()
        again = g.read_framebuffer()         # gathering twice does not disturb the members' running means
        gc = g.counters()
    finally:
        g.close()
    err = np.abs(img[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-3)
    assert err.max() < 2e-6                  # summation order differs from the sequential running mean
    assert (img[..., 3] == 1.0).all()
    assert np.array_equal(img, again)
    assert gc["rays_primary"] == rc["rays_primary"] and gc["rays_shadow"] == rc["rays_shadow"]


def test_group_argument_checks(pt):
    with pytest.raises(pt.PtError, match="empty device list"):
        pt.Group(W, H, [], mode=pt.PT_GROUP_TILES)
    with pytest.raises(pt.PtError, match="unknown mode"):
        pt.Group(W, H, [0], mode=7)
    with pytest.raises(pt.PtError, match="device"):
        pt.Group(W, H, [0, 99], mode=pt.PT_GROUP_TILES)
    g = pt.Group(W, H, [0, 0])
    try:
        with pytest.raises(pt.PtError, match="upload a scene"):
            g.render(0, pt.frame_seeds(0, 1), 5)
    finally:
        g.close()


def test_cpp_main_on_a_device_group(pt, tmp_path):
    """tools/pt355_main.cpp with a device list: the reference's main() sequence on a pt_group (three virtual devices on
    the one GPU), tiles -> bit-identical to its own single-device run."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pt.__file__), "pt355_main")
    one, many = str(tmp_path / "one.pfm"), str(tmp_path / "many.pfm")
    for pfm, extra in ((one, []), (many, ["-", "0,0,0", "tiles"])):
        out = subprocess.run([exe, "4", "256", "144", str(tmp_path / "x.ppm"), pfm] + extra, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stderr + out.stdout
    a, b = pt.read_pfm(one, 256, 144), pt.read_pfm(many, 256, 144)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
