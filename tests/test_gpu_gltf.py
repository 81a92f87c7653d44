"""A glTF scene end to end on the GPU: stand-in atrium -> .glb (tools/make_scene_glb.py) -> loader -> scene set-up of
src/main.cpp -> upload with its texture array and sampler state -> render; bit-identical to the oracle on the same arrays."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_textured_gltf_scene_renders_like_the_oracle(pt, ol, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_scene_glb
    tex = pt.make_textures(6, 1)
    path = str(tmp_path / "atrium1.glb")
    n_prims = make_scene_glb.write_scene_glb(pt.Scene.atrium(detail=1, seed=1), path, tex)
    sc = pt.Scene.from_gltf(path)
    d = sc.desc()
    assert d.n_materials == n_prims + 4 and d.n_light_tris > 0 and d.n_spheres == 4
    assert sc.textures is not None and sc.textures.shape[0] >= 6 and sc.tex_sampler == (10497, 10497, 9729)
    cam, seeds = pt.default_camera(), pt.frame_seeds(0, 3)
    c = pt.PathTracer(160, 96)
    try:
        c.upload_scene(sc, tex_rgb8=sc.textures, tex_sampler=sc.tex_sampler)
        c.set_camera(cam)
        c.render(0, seeds, 5)
        img = c.read_framebuffer()
        cnt = c.counters()
    finally:
        c.close()
    ref, ocnt = ol.render(ol.scene_from_desc(sc.desc(), tex_rgb8=sc.textures, tex_sampler=sc.tex_sampler), cam, 160, 96, seeds, bounces=5)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert cnt["rays_primary"] == ocnt["rays_primary"]
    assert img[..., :3].max() > 0.05
