"""A lit box with TEXTURED sphere materials in front of triangles that carry uvs.

shaders/framebuffer.comp:447-451 (the sphere branch of intersect) writes t, normal, mat_index, object_index and
object_type but not `uvs`: a sphere hit keeps the uvs of the last triangle intersect_bvh_stack accepted, and calc_BRDF
(fc:599-603) samples the sphere material's texture layer with them.  Every sphere here has a textured material, one per
BRDF family, and sits in front of a wall whose triangles have non-trivial uvs -- textured walls (floor, back) and
untextured ones (sides, ceiling: their uvs matter ONLY through the spheres in front of them; `wall_uv_shift` moves
them)."""
import numpy as np


def build(pt, wall_uv_shift=0.0, extra=()):
    """extra: more spheres as (origin, radius, mat_type, roughness, tex_index) -- the soak draws them at random."""
    s = pt.Scene()
    white = s.add_material([0.73, 0.73, 0.73], [0, 0, 0], [0, 0, 0])
    floor = s.add_material([0.9, 0.9, 0.9], [0, 0, 0], [0, 0, 0], roughness=0.2, tex_index=0, mat_type=pt.MAT_OREN_NAYAR)
    back = s.add_material([1.0, 1.0, 1.0], [0, 0, 0], [0, 0, 0], tex_index=1, mat_type=pt.MAT_LAMBERTIAN)
    light = s.add_material([0, 0, 0], [0, 0, 0], [17, 12, 4], mat_type=pt.MAT_LIGHT)
    # the spheres' materials: all three BRDF families, each sampling a texture layer
    sph_lambert = s.add_material([1.0, 0.9, 0.8], [0, 0, 0], [0, 0, 0], tex_index=1, mat_type=pt.MAT_LAMBERTIAN)
    sph_oren = s.add_material([0.9, 1.0, 0.9], [0, 0, 0], [0, 0, 0], roughness=0.3, tex_index=0, mat_type=pt.MAT_OREN_NAYAR)
    sph_metal = s.add_material([0, 0, 0], [0.944, 0.776, 0.373], [0, 0, 0], roughness=0.15, tex_index=2, mat_type=pt.MAT_SPECULAR_METAL)
    sph_light = s.add_material([0, 0, 0], [0, 0, 0], [6, 6, 9], tex_index=2, mat_type=pt.MAT_LIGHT)

    def quad(a, b, c, d, mat, uvscale=1.0, shift=0.0):
        a, b, c, d = (np.array(p, np.float32) for p in (a, b, c, d))
        n = np.cross(b - a, c - a)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uvscale - 0.25 + np.float32(shift)
        v = np.stack([np.concatenate([a, b, c]), np.concatenate([a, c, d])])
        nn = np.tile(n, (2, 3))
        uvs = np.stack([np.concatenate([uv[0], uv[1], uv[2]]), np.concatenate([uv[0], uv[2], uv[3]])])
        s.add_triangles(v, nn, [mat, mat], uvs)

    x0, x1, y0, y1, z0, z1 = -2.0, 2.0, -2.0, 2.0, -8.0, -4.0
    quad([x0, y0, z1], [x1, y0, z1], [x1, y0, z0], [x0, y0, z0], floor, 3.0)
    quad([x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], back, 2.0)
    quad([x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1], white, 1.7, wall_uv_shift)
    quad([x0, y0, z1], [x0, y0, z0], [x0, y1, z0], [x0, y1, z1], white, 2.3, wall_uv_shift)
    quad([x1, y0, z0], [x1, y0, z1], [x1, y1, z1], [x1, y1, z0], white, 1.3, wall_uv_shift)
    quad([-0.5, 1.99, -6.5], [0.5, 1.99, -6.5], [0.5, 1.99, -5.5], [-0.5, 1.99, -5.5], light)
    s.add_sphere([0.0, 0.2, -6.8], 0.7, sph_lambert)    # in front of the textured back wall
    s.add_sphere([-1.3, -0.3, -5.6], 0.5, sph_oren)      # in front of the untextured left wall
    s.add_sphere([1.2, -1.3, -5.4], 0.6, sph_metal)      # above the textured floor, beside the right wall
    s.add_sphere([1.3, 1.2, -6.6], 0.25, sph_light)      # an emissive sphere (a sphere light for NEE)
    for (o, r, mt, rough, ti) in extra:
        spec = [0.944, 0.776, 0.373] if mt == pt.MAT_SPECULAR_METAL else [0, 0, 0]
        s.add_sphere(list(o), float(r), s.add_material([0.8, 0.8, 0.8], spec, [0, 0, 0], roughness=float(rough), tex_index=int(ti), mat_type=int(mt)))
    s.finalize()
    rng = np.random.default_rng(12)
    tex = rng.integers(0, 256, (3, 512, 512, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:512, 0:512]
    tex[0] = np.stack([((xx // 32 + yy // 32) % 2) * 200 + 30, xx // 2, yy // 2], 2).astype(np.uint8)  # checker + ramps
    tex[1] = np.stack([xx // 2, ((xx // 16 + yy // 48) % 2) * 180 + 60, 255 - yy // 2], 2).astype(np.uint8)
    return s, tex


SPHERES = [((0.0, 0.2, -6.8), 0.7), ((-1.3, -0.3, -5.6), 0.5), ((1.2, -1.3, -5.4), 0.6), ((1.3, 1.2, -6.6), 0.25)]
