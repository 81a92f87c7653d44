"""pathtracer_amd -- MI355X-native render hot path of limepixl/pathtracer behind a C ABI.

This package is a thin ctypes view of ``libpt355.so`` (include/pt355.h, include/pt355_host.h).
There is no Python or CPU implementation of the path behind it: if the library is missing, or no
HIP device is present when a context is created, the call fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT355_LIB") or os.path.join(_HERE, "libpt355.so")  # PT355_LIB: an A/B build of the same library (tools/exp_build.sh)

PT_FLAG_SHADOW_ANYHIT = 1
PT_FLAG_TRACE_DEAD_PATHS = 2
PT_FLAG_COUNT_TRAVERSAL = 4
PT_BVH_PLOC = 0x100  # or-ed into max_leaf of the GPU BVH builders: clustering (PLOC) instead of the radix tree
PT_FLAG_SHADOW_FULL_TRAVERSAL = 8
PT_ESTIMATOR_MIS, PT_ESTIMATOR_NEE, PT_ESTIMATOR_BRDF = 0, 1, 2
PT_PARTITION_TILES = 0
PT_PARTITION_FULL = 1
MAT_LIGHT, MAT_LAMBERTIAN, MAT_OREN_NAYAR, MAT_SPECULAR_METAL = -1, 0, 1, 2


class PtError(RuntimeError):
    pass


class pt_config(ctypes.Structure):
    _fields_ = [("width", c_uint32), ("height", c_uint32), ("device", c_int32), ("flags", c_uint32),
                ("partition", c_uint32), ("rank", c_uint32), ("world", c_uint32), ("max_batch", c_uint32), ("pipelines", c_uint32), ("reserved", c_uint32 * 7)]


class pt_scene_desc(ctypes.Structure):
    _fields_ = [("spheres", c_void_p), ("n_spheres", c_uint32),
                ("tris", c_void_p), ("n_tris", c_uint32),
                ("light_tris", c_void_p), ("n_light_tris", c_uint32),
                ("materials", c_void_p), ("n_materials", c_uint32),
                ("nodes", c_void_p), ("n_nodes", c_uint32),
                ("light_spheres", c_void_p), ("n_light_spheres", c_uint32),
                ("env_rgb8", c_void_p), ("env_w", c_uint32), ("env_h", c_uint32),
                ("tex_rgb8", c_void_p), ("tex_layers", c_uint32),
                ("tex_wrap_s", c_uint32), ("tex_wrap_t", c_uint32), ("tex_mag_filter", c_uint32)]


class pt_counters(ctypes.Structure):
    _fields_ = [("frames", c_uint64), ("paths", c_uint64), ("rays_primary", c_uint64), ("rays_shadow", c_uint64),
                ("rays_bounce", c_uint64), ("node_steps", c_uint64), ("tri_tests", c_uint64), ("stack_spills", c_uint64),
                ("ms_trace", c_double), ("ms_shade", c_double), ("trace_launches", c_uint64), ("wave_iters", c_uint64), ("top_steps", c_uint64), ("reserved", c_uint64 * 3)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


# every symbol include/pt355.h and include/pt355_host.h declare (tests check that all of them resolve)
EXPORTS = {
    "pt_create": (c_int, [POINTER(pt_config), POINTER(c_void_p)]),
    "pt_destroy": (c_int, [c_void_p]),
    "pt_last_error": (c_char_p, [c_void_p]),
    "pt_upload_scene": (c_int, [c_void_p, POINTER(pt_scene_desc)]),
    "pt_set_camera": (c_int, [c_void_p, c_void_p]),
    "pt_set_estimator": (c_int, [c_void_p, c_uint32]),
    "pt_render_frame": (c_int, [c_void_p, c_void_p]),
    "pt_render": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32]),
    "pt_sync": (c_int, [c_void_p]),
    "pt_read_framebuffer": (c_int, [c_void_p, c_void_p, c_size_t]),
    "pt_clear_framebuffer": (c_int, [c_void_p]),
    "pt_write_framebuffer": (c_int, [c_void_p, c_void_p, c_size_t]),
    "pt_framebuffer_device": (c_int, [c_void_p, POINTER(c_void_p)]),
    "pt_trace_closest": (c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pt_trace_closest_device": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p]),
    "pt_trace_closest_device_noalloc": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p]),
    "pt_get_info": (c_int, [c_void_p, c_void_p]),
    "pt_get_counters": (c_int, [c_void_p, POINTER(pt_counters)]),
    "pt_reset_counters": (c_int, [c_void_p]),
    "pt_partition_count": (c_uint32, [c_uint32, c_uint32, c_uint32, c_uint32, c_uint32]),
    "pt_partition_pixels": (c_int, [c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_void_p]),
    "pt_pack_owned": (c_int, [c_void_p, c_void_p]),
    "pt_unpack_from": (c_int, [c_void_p, c_void_p, c_uint32, c_uint32]),
    "pt_debug_eval": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_uint32, c_void_p]),
    "pt_debug_read_queue": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_size_t, POINTER(c_uint32), POINTER(c_uint32)]),
    "pt_build_bvh_device": (c_int, [c_void_p, c_void_p, c_uint32, c_uint32, c_void_p, POINTER(c_uint32), c_void_p, POINTER(c_float)]),
    "pt_scene_finalize_device": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_float)]),
    "pt_group_create": (c_int, [POINTER(pt_config), c_void_p, c_uint32, c_uint32, POINTER(c_void_p)]),
    "pt_group_destroy": (c_int, [c_void_p]),
    "pt_group_last_error": (c_char_p, [c_void_p]),
    "pt_group_size": (c_uint32, [c_void_p]),
    "pt_group_member": (c_void_p, [c_void_p, c_uint32]),
    "pt_group_upload_scene": (c_int, [c_void_p, POINTER(pt_scene_desc)]),
    "pt_group_set_camera": (c_int, [c_void_p, c_void_p]),
    "pt_group_set_estimator": (c_int, [c_void_p, c_uint32]),
    "pt_group_render": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_uint32]),
    "pt_group_sync": (c_int, [c_void_p]),
    "pt_group_clear": (c_int, [c_void_p]),
    "pt_group_gather": (c_int, [c_void_p, POINTER(c_void_p)]),
    "pt_group_read_framebuffer": (c_int, [c_void_p, c_void_p, c_size_t]),
    "pt_group_get_counters": (c_int, [c_void_p, POINTER(pt_counters)]),
    # host side
    "pth_pack_half2x16": (c_uint32, [c_float, c_float]),
    "pth_oct_encode": (None, [c_void_p, c_void_p]),
    "pth_pack_triangle": (None, [c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "pth_pack_material": (None, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p]),
    "pth_pack_sphere": (None, [c_void_p, c_float, c_uint32, c_void_p]),
    "pth_pack_camera": (None, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p]),
    "pth_build_bvh": (c_uint32, [c_void_p, c_uint32, c_void_p, c_void_p]),
    "pth_find_emissive_tris": (c_uint32, [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p]),
    "pth_find_emissive_spheres": (c_uint32, [c_void_p, c_uint32, c_void_p, c_uint32, c_void_p]),
    "pth_frame_seeds": (None, [c_uint32, c_uint32, c_void_p]),
    "pth_scene_create": (c_void_p, []),
    "pth_scene_destroy": (None, [c_void_p]),
    "pth_scene_add_material": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int]),
    "pth_scene_add_sphere": (c_int, [c_void_p, c_void_p, c_float, c_uint32]),
    "pth_scene_add_triangles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    "pth_scene_fat_count": (c_uint32, [c_void_p]),
    "pth_scene_get_fat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pth_scene_finalize": (c_int, [c_void_p]),
    "pth_scene_finalize_with": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pth_scene_desc": (c_int, [c_void_p, POINTER(pt_scene_desc)]),
    "pth_scene_hash": (c_uint64, [c_void_p]),
    "pth_scene_make_cornell": (c_int, [c_void_p, c_uint32]),
    "pth_scene_make_bunny": (c_int, [c_void_p, c_uint32, c_uint32, c_uint32, c_int, c_float, c_uint32]),
    "pth_scene_make_atrium": (c_int, [c_void_p, c_uint32, c_uint32]),
    "pth_make_textures": (c_int, [c_uint32, c_uint32, c_void_p]),
    "pth_scene_make_soup": (c_int, [c_void_p, c_uint32, c_uint32, c_float]),
    "pth_scene_add_packed_materials": (c_uint32, [c_void_p, c_void_p, c_uint32]),
    "pth_model_create": (c_void_p, []),
    "pth_model_destroy": (None, [c_void_p]),
    "pth_last_error": (c_char_p, []),
    "pth_load_gltf": (c_int, [c_char_p, c_void_p]),
    "pth_model_translate": (None, [c_void_p, c_void_p]),
    "pth_model_rotate": (None, [c_void_p, c_void_p]),
    "pth_model_scale": (None, [c_void_p, c_void_p]),
    "pth_model_apply_matrix": (None, [c_void_p]),
    "pth_model_triangle_count": (c_uint32, [c_void_p]),
    "pth_model_material_count": (c_uint32, [c_void_p]),
    "pth_model_get": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pth_model_textures": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_uint32), c_void_p]),
    "pth_scene_add_model": (c_int, [c_void_p, c_void_p]),
    "pth_load_env_hdr": (c_int, [c_char_p, c_void_p, c_size_t, POINTER(c_uint32), POINTER(c_uint32)]),
    "pth_decode_image_rgb8": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "pth_resize_rgb8_512": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "pth_tonemap_rgb8": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p]),
    "pth_write_ppm": (c_int, [c_char_p, c_void_p, c_uint32, c_uint32]),
    "pth_write_pfm": (c_int, [c_char_p, c_void_p, c_uint32, c_uint32]),
    "pth_read_pfm": (c_int, [c_char_p, c_void_p, c_uint32, c_uint32]),
}

_lib = None

KERNEL_SOURCES = ("device_math.h", "kernels_trace.h", "kernels_shade.h")


def kernel_source_sha256() -> str:
    """SHA-256 over the device code of the hot path (csrc/device_math.h, kernels_trace.h, kernels_shade.h, in that order).
    tools/prof_summary.py stamps every profiles/*_pmc_traffic.json with it and bench.py refuses to price a roofline with PMC
    figures whose stamp is not that of the sources it runs: a profile of another kernel is not evidence for this one."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def lib():
    """Load libpt355.so once.  torch is imported first when available so that the HIP runtime the
    library binds to is the one torch (and RCCL) already loaded -- two HIP runtimes in one process
    cannot share device pointers."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                      "(or python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
    if os.environ.get("PT355_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


GL_REPEAT, GL_CLAMP_TO_EDGE, GL_MIRRORED_REPEAT, GL_NEAREST, GL_LINEAR = 10497, 33071, 33648, 9728, 9729


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def frame_seeds(first: int, n: int) -> np.ndarray:
    """pcg32_random() outputs number first..first+n-1 of the default global stream (src/main.cpp:97)."""
    out = np.zeros(n, np.uint32)
    lib().pth_frame_seeds(first, n, _ptr(out))
    return out


def default_camera() -> np.ndarray:
    """Camera of src/main.cpp:67 as CameraGLSL (12 floats)."""
    cam = np.zeros(12, np.float32)
    o = np.zeros(3, np.float32)
    f = np.array([0, 0, -1], np.float32)
    r = np.array([1, 0, 0], np.float32)
    lib().pth_pack_camera(_ptr(o), _ptr(f), _ptr(r), 0.005, 0.05, _ptr(cam))
    return cam


class Model:
    """LoadGLTF + Model (src/loader.h:4, src/scene/model.h:7-29)."""

    def __init__(self, path: str | None = None):
        self._h = lib().pth_model_create()
        if path is not None:
            self.load(path)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pth_model_destroy(self._h)
            self._h = None

    def load(self, path: str):
        if lib().pth_load_gltf(os.fsencode(path), self._h) != 0:
            raise PtError(lib().pth_last_error().decode())
        return self

    def translate(self, t):
        lib().pth_model_translate(self._h, _ptr(np.asarray(t, np.float32)))

    def rotate(self, r):
        lib().pth_model_rotate(self._h, _ptr(np.asarray(r, np.float32)))

    def scale(self, s):
        s = np.full(3, s, np.float32) if np.isscalar(s) else np.asarray(s, np.float32)
        lib().pth_model_scale(self._h, _ptr(s))

    def apply_matrix(self):
        lib().pth_model_apply_matrix(self._h)

    def arrays(self) -> dict:
        n, nm = lib().pth_model_triangle_count(self._h), lib().pth_model_material_count(self._h)
        v, nn, uv = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32), np.zeros((n, 6), np.float32)
        mat, mats = np.zeros(n, np.uint32), np.zeros((nm, 12), np.uint32)
        lib().pth_model_get(self._h, _ptr(v), _ptr(nn), _ptr(uv), _ptr(mat), _ptr(mats))
        return {"v": v, "n": nn, "uv": uv, "mat": mat, "materials": mats}

    def textures(self):
        """(u8 [layers, 512, 512, 3] or None, (wrap_s, wrap_t, mag_filter))"""
        p, layers, smp = c_void_p(), c_uint32(), np.zeros(3, np.uint32)
        lib().pth_model_textures(self._h, byref(p), byref(layers), _ptr(smp))
        tex = None
        if p.value and layers.value:
            tex = np.ctypeslib.as_array(ctypes.cast(p, POINTER(c_uint8)), shape=(layers.value, 512, 512, 3)).copy()
        return tex, tuple(int(x) for x in smp)


class Scene:
    """Host-side scene container (src/main.cpp:18-63): triangles, materials, spheres -> six arrays."""

    def __init__(self):
        self._h = lib().pth_scene_create()
        self._keep = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pth_scene_destroy(self._h)
            self._h = None

    @classmethod
    def cornell(cls, spheres: bool = False) -> "Scene":
        s = cls()
        _check_host(lib().pth_scene_make_cornell(s._h, 1 if spheres else 0), "pth_scene_make_cornell")
        return s

    @classmethod
    def bunny(cls, freq: int = 59, target_tris: int = 69451, seed: int = 1, material: int = MAT_OREN_NAYAR,
              roughness: float = 0.175, spheres: bool = False) -> "Scene":
        s = cls()
        _check_host(lib().pth_scene_make_bunny(s._h, freq, target_tris, seed, material, roughness, 1 if spheres else 0),
                    "pth_scene_make_bunny")
        return s

    @classmethod
    def soup(cls, n_tris: int, seed: int = 7, size: float = 0.25) -> "Scene":
        s = cls()
        _check_host(lib().pth_scene_make_soup(s._h, n_tris, seed, size), "pth_scene_make_soup")
        return s

    @classmethod
    def atrium(cls, detail: int = 4, seed: int = 1) -> "Scene":
        """Sponza-class stand-in: ~17,600 * detail^2 triangles, 27 materials, texture layers 0..5."""
        s = cls()
        _check_host(lib().pth_scene_make_atrium(s._h, detail, seed), "pth_scene_make_atrium")
        return s

    @classmethod
    def from_gltf(cls, path: str, main_setup: bool = True, spheres: bool = True) -> "Scene":
        """The scene src/main.cpp:18-55 builds around a model file: LoadGLTF, Translate(0,-2,-6) . Rotate(0,-90 deg,0) .
        Scale(2) baked into the triangles, the four gold GGX spheres, CalculateBVH, emissive lists.  The texture array
        and its sampler state are kept on the object (`textures`, `tex_sampler`) for PathTracer.upload_scene."""
        m = Model(path)
        if main_setup:
            m.translate([0.0, -2.0, -6.0])
            m.rotate([0.0, np.float32(np.deg2rad(np.float32(-90.0))), 0.0])
            m.scale(2.0)
            m.apply_matrix()
        s = cls()
        s.add_model(m)
        if spheres:
            for rough, cx in ((0.0, -1.0), (0.1, -0.4), (0.15, 0.2), (0.2, 0.8)):
                mi = s.add_material([0, 0, 0], [0.944, 0.776, 0.373], [0, 0, 0], rough, -1, MAT_SPECULAR_METAL)
                s.add_sphere([cx, 1.0, -5.0], 0.3, mi)
        s.finalize()
        s.textures, smp = m.textures()
        s.tex_sampler = smp if s.textures is not None else None
        return s

    def add_model(self, model: "Model"):
        _check_host(lib().pth_scene_add_model(self._h, model._h), "pth_scene_add_model")

    def add_material(self, diffuse, specular, le, roughness=0.0, tex_index=-1, mat_type=MAT_LAMBERTIAN) -> int:
        d, s, l = (np.asarray(x, np.float32) for x in (diffuse, specular, le))
        return lib().pth_scene_add_material(self._h, _ptr(d), _ptr(s), _ptr(l), roughness, tex_index, mat_type)

    def add_sphere(self, origin, radius, mat_index) -> int:
        o = np.asarray(origin, np.float32)
        return lib().pth_scene_add_sphere(self._h, _ptr(o), radius, mat_index)

    def add_triangles(self, v, n, mat, uv=None):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 9)
        n = np.ascontiguousarray(n, np.float32).reshape(-1, 9)
        mat = np.ascontiguousarray(mat, np.uint32).reshape(-1)
        uvp = None if uv is None else np.ascontiguousarray(uv, np.float32).reshape(-1, 6)
        _check_host(lib().pth_scene_add_triangles(self._h, _ptr(v), _ptr(n), _ptr(uvp), _ptr(mat), v.shape[0]),
                    "pth_scene_add_triangles")

    def finalize(self):
        _check_host(lib().pth_scene_finalize(self._h), "pth_scene_finalize")

    def fat_triangles(self) -> dict:
        """The triangles as they were added (v [n,9], n [n,9], uv [n,6], mat [n]), before packing and BVH order."""
        n = lib().pth_scene_fat_count(self._h)
        v, nn, uv, mat = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32), np.zeros((n, 6), np.float32), np.zeros(n, np.uint32)
        lib().pth_scene_get_fat(self._h, _ptr(v), _ptr(nn), _ptr(uv), _ptr(mat))
        return {"v": v, "n": nn, "uv": uv, "mat": mat}

    def desc(self) -> pt_scene_desc:
        d = pt_scene_desc()
        _check_host(lib().pth_scene_desc(self._h, byref(d)), "pth_scene_desc")
        return d

    def hash(self) -> int:
        return lib().pth_scene_hash(self._h)

    def arrays(self) -> dict:
        """Copies of the six arrays as numpy (u32 views of the raw records)."""
        d = self.desc()

        def grab(ptr, n, words):
            if not ptr or n == 0:
                return np.zeros((0, words), np.uint32)
            return np.ctypeslib.as_array(ctypes.cast(ptr, POINTER(c_uint32)), shape=(n, words)).copy()

        return {"spheres": grab(d.spheres, d.n_spheres, 8), "tris": grab(d.tris, d.n_tris, 16),
                "light_tris": grab(d.light_tris, d.n_light_tris, 1).reshape(-1),
                "materials": grab(d.materials, d.n_materials, 12), "nodes": grab(d.nodes, d.n_nodes, 8),
                "light_spheres": grab(d.light_spheres, d.n_light_spheres, 1).reshape(-1)}


def _check_host(rc, what):
    if rc != 0:
        raise PtError(f"{what} failed with code {rc}")


class PathTracer:
    """One pt_ctx: the replacement for the reference's GL upload + glDispatchCompute sequence."""

    def __init__(self, width: int, height: int, device: int = 0, flags: int = 0, partition: int = PT_PARTITION_FULL,
                 rank: int = 0, world: int = 1, max_batch: int = 0, pipelines: int = 0):
        L = lib()
        cfg = pt_config(width=width, height=height, device=device, flags=flags, partition=partition, rank=rank, world=world,
                        max_batch=max_batch, pipelines=pipelines)
        h = c_void_p()
        rc = L.pt_create(byref(cfg), byref(h))
        if rc != 0:
            raise PtError(f"pt_create: {L.pt_last_error(None).decode()}")
        self._h = h
        self.width, self.height = width, height
        self.rank, self.world, self.partition = rank, world, partition
        info = np.zeros(8, np.uint32)
        self._check(L.pt_get_info(h, _ptr(info)), "pt_get_info")
        self.max_batch, self.pipelines, self.queue_stride = int(info[0]), int(info[1]), int(info[2])
        self.info = info  # pt_get_info: [5] = k_trace grid (blocks)
        self._scene = None

    def _check(self, rc, what):
        if rc != 0:
            raise PtError(f"{what}: {lib().pt_last_error(self._h).decode()} (code {rc})")

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def upload_scene(self, scene, env_rgb8: np.ndarray | None = None, tex_rgb8: np.ndarray | None = None,
                     tex_sampler: tuple | None = None):
        """tex_sampler = (wrap_s, wrap_t, mag_filter) as GL enums (GL_REPEAT ... below); None = the loader's defaults."""
        d = scene.desc() if isinstance(scene, Scene) else scene
        if tex_sampler is not None:
            d.tex_wrap_s, d.tex_wrap_t, d.tex_mag_filter = tex_sampler
        keep = [scene]
        if env_rgb8 is not None:
            env = np.ascontiguousarray(env_rgb8, np.uint8)
            if env.ndim != 3 or env.shape[2] != 3 or np.asarray(env_rgb8).dtype != np.uint8:
                raise PtError(f"upload_scene: env_rgb8 must be uint8 [H, W, 3], got {np.asarray(env_rgb8).dtype} {env.shape}")
            d.env_rgb8, d.env_h, d.env_w = env.ctypes.data, env.shape[0], env.shape[1]
            keep.append(env)
        if tex_rgb8 is not None:
            tex = np.ascontiguousarray(tex_rgb8, np.uint8)
            if tex.ndim != 4 or tex.shape[1:] != (512, 512, 3) or np.asarray(tex_rgb8).dtype != np.uint8:
                raise PtError(f"upload_scene: tex_rgb8 must be uint8 [layers, 512, 512, 3], got {np.asarray(tex_rgb8).dtype} {tex.shape}")
            d.tex_rgb8, d.tex_layers = tex.ctypes.data, tex.shape[0]
            keep.append(tex)
        self._check(lib().pt_upload_scene(self._h, byref(d)), "pt_upload_scene")
        self._scene = keep

    def set_camera(self, cam12):
        cam = np.ascontiguousarray(cam12, np.float32).reshape(12)
        self._check(lib().pt_set_camera(self._h, _ptr(cam)), "pt_set_camera")

    def set_estimator(self, which: int):
        self._check(lib().pt_set_estimator(self._h, which), "pt_set_estimator")

    def render_frame(self, seed: int, frame: int, bounces: int = 5):
        fd = np.array([seed, frame, bounces, 0], np.uint32)
        self._check(lib().pt_render_frame(self._h, _ptr(fd)), "pt_render_frame")

    def render(self, first_frame: int, seeds, bounces: int = 5):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        self._check(lib().pt_render(self._h, first_frame, seeds.size, _ptr(seeds), bounces), "pt_render")

    def sync(self):
        self._check(lib().pt_sync(self._h), "pt_sync")

    def read_framebuffer(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib().pt_read_framebuffer(self._h, _ptr(out), out.size), "pt_read_framebuffer")
        return out

    def write_framebuffer(self, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        self._check(lib().pt_write_framebuffer(self._h, _ptr(a), a.size), "pt_write_framebuffer")

    def clear_framebuffer(self):
        self._check(lib().pt_clear_framebuffer(self._h), "pt_clear_framebuffer")

    def framebuffer_device(self) -> int:
        p = c_void_p()
        self._check(lib().pt_framebuffer_device(self._h, byref(p)), "pt_framebuffer_device")
        return p.value

    def trace_closest(self, ro, rd):
        ro = np.ascontiguousarray(ro, np.float32).reshape(-1, 3)
        rd = np.ascontiguousarray(rd, np.float32).reshape(-1, 3)
        n = ro.shape[0]
        prim = np.zeros(n, np.uint32)
        t_h = np.zeros(n, np.uint16)
        kind = np.zeros(n, np.uint8)
        uv = np.zeros((n, 2), np.float32)
        self._check(lib().pt_trace_closest(self._h, _ptr(ro), _ptr(rd), n, _ptr(prim), _ptr(t_h), _ptr(kind), _ptr(uv)),
                    "pt_trace_closest")
        return prim, t_h, kind, uv

    def trace_closest_device(self, ray6_ptr: int, n: int, hit4_ptr: int):
        self._check(lib().pt_trace_closest_device(self._h, c_void_p(ray6_ptr), n, c_void_p(hit4_ptr)), "pt_trace_closest_device")

    def trace_closest_device_noalloc(self, ray6_ptr: int, n: int, hit4_ptr: int):
        self._check(lib().pt_trace_closest_device_noalloc(self._h, c_void_p(ray6_ptr), n, c_void_p(hit4_ptr)),
                    "pt_trace_closest_device_noalloc")

    def counters(self) -> dict:
        c = pt_counters()
        self._check(lib().pt_get_counters(self._h, byref(c)), "pt_get_counters")
        return c.as_dict()

    def reset_counters(self):
        self._check(lib().pt_reset_counters(self._h), "pt_reset_counters")

    def pack_owned(self, dst_ptr: int):
        self._check(lib().pt_pack_owned(self._h, c_void_p(dst_ptr)), "pt_pack_owned")

    def unpack_from(self, src_ptr: int, from_rank: int, world: int):
        self._check(lib().pt_unpack_from(self._h, c_void_p(src_ptr), from_rank, world), "pt_unpack_from")

    def build_bvh_device(self, tris, max_leaf: int = 4, ploc: bool = False):
        """GPU BVH (LBVH, or PLOC with ploc=True) over n packed 64-byte triangles (u32 [n,16]); returns (nodes u32 [m,8],
        sorted tris u32 [n,16], build ms)."""
        tris = np.ascontiguousarray(tris, np.uint32).reshape(-1, 16)
        n = tris.shape[0]
        nodes = np.zeros((2 * n + 1, 8), np.uint32)
        out = np.zeros_like(tris)
        nn, ms = c_uint32(), c_float()
        self._check(lib().pt_build_bvh_device(self._h, _ptr(tris), n, max_leaf | (PT_BVH_PLOC if ploc else 0), _ptr(nodes), byref(nn), _ptr(out),
                                              byref(ms)), "pt_build_bvh_device")
        return nodes[: nn.value].copy(), out, ms.value

    def finalize_scene_device(self, scene: "Scene", max_leaf: int = 4, ploc: bool = False) -> float:
        """Re-finalizes `scene` with the GPU builder in place of CalculateBVH; returns the build's device ms."""
        ms = c_float()
        self._check(lib().pt_scene_finalize_device(self._h, scene._h, max_leaf | (PT_BVH_PLOC if ploc else 0), byref(ms)), "pt_scene_finalize_device")
        return ms.value

    def debug_read_queue(self, queue: int, field: int) -> np.ndarray:
        cols = [1, 1, 1, 3, 3, 6, 6, 8, 5, 4, 4][field]
        out = np.zeros((cols, self.queue_stride), np.uint32)
        st, cnt = c_uint32(), c_uint32()
        self._check(lib().pt_debug_read_queue(self._h, queue, field, _ptr(out), out.size, byref(st), byref(cnt)), "pt_debug_read_queue")
        return out, cnt.value

    def debug_eval(self, op: int, a, b=None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        bb = None if b is None else np.ascontiguousarray(b, np.float32)
        out = np.empty_like(a)
        self._check(lib().pt_debug_eval(self._h, op, _ptr(a), _ptr(bb), a.size, _ptr(out)), "pt_debug_eval")
        return out


PT_GROUP_TILES, PT_GROUP_SAMPLES = 0, 1


class Group:
    """pt_group: several devices (or several virtual devices on one) behind one handle, in one process."""

    def __init__(self, width: int, height: int, devices, mode: int = PT_GROUP_TILES, flags: int = 0, max_batch: int = 0):
        L = lib()
        cfg = pt_config(width=width, height=height, flags=flags, max_batch=max_batch)
        dev = np.ascontiguousarray(devices, np.int32)
        h = c_void_p()
        if L.pt_group_create(byref(cfg), _ptr(dev), dev.size, mode, byref(h)) != 0:
            raise PtError(f"pt_group_create: {L.pt_group_last_error(None).decode()}")
        self._h, self.width, self.height, self.mode = h, width, height, mode

    def _check(self, rc, what):
        if rc != 0:
            raise PtError(f"{what}: {lib().pt_group_last_error(self._h).decode()} (code {rc})")

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def upload_scene(self, scene, tex_rgb8=None, tex_sampler=None):
        d = scene.desc() if isinstance(scene, Scene) else scene
        self._keep = [scene]
        if tex_rgb8 is not None:
            tex = np.ascontiguousarray(tex_rgb8, np.uint8)
            d.tex_rgb8, d.tex_layers = tex.ctypes.data, tex.shape[0]
            self._keep.append(tex)
        if tex_sampler is not None:
            d.tex_wrap_s, d.tex_wrap_t, d.tex_mag_filter = tex_sampler
        self._check(lib().pt_group_upload_scene(self._h, byref(d)), "pt_group_upload_scene")

    def set_camera(self, cam12):
        cam = np.ascontiguousarray(cam12, np.float32).reshape(12)
        self._check(lib().pt_group_set_camera(self._h, _ptr(cam)), "pt_group_set_camera")

    def render(self, first_frame: int, seeds, bounces: int = 5):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        self._check(lib().pt_group_render(self._h, first_frame, seeds.size, _ptr(seeds), bounces), "pt_group_render")

    def sync(self):
        self._check(lib().pt_group_sync(self._h), "pt_group_sync")

    def clear(self):
        self._check(lib().pt_group_clear(self._h), "pt_group_clear")

    def read_framebuffer(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib().pt_group_read_framebuffer(self._h, _ptr(out), out.size), "pt_group_read_framebuffer")
        return out

    def counters(self) -> dict:
        c = pt_counters()
        self._check(lib().pt_group_get_counters(self._h, byref(c)), "pt_group_get_counters")
        return c.as_dict()


def load_env_hdr(path: str) -> np.ndarray:
    """Radiance .hdr -> the 8-bit RGB image Display::InitRenderBuffer uploads as the env map (u8 [h, w, 3])."""
    w, h = c_uint32(), c_uint32()
    if lib().pth_load_env_hdr(os.fsencode(path), None, 0, byref(w), byref(h)) != 0:
        raise PtError(lib().pth_last_error().decode())
    out = np.zeros((h.value, w.value, 3), np.uint8)
    if lib().pth_load_env_hdr(os.fsencode(path), _ptr(out), out.size, None, None) != 0:
        raise PtError(lib().pth_last_error().decode())
    return out


def make_textures(layers: int = 6, seed: int = 1) -> np.ndarray:
    """Procedural 512x512 RGB8 layers for Scene.atrium (u8 [layers, 512, 512, 3])."""
    out = np.zeros((layers, 512, 512, 3), np.uint8)
    _check_host(lib().pth_make_textures(layers, seed, _ptr(out)), "pth_make_textures")
    return out


def partition_pixels(width: int, height: int, partition: int, rank: int, world: int) -> np.ndarray:
    n = lib().pt_partition_count(width, height, partition, rank, world)
    out = np.zeros(n, np.uint32)
    if n:
        _check_host(lib().pt_partition_pixels(width, height, partition, rank, world, _ptr(out)), "pt_partition_pixels")
    return out


def tonemap_rgb8(rgba: np.ndarray) -> np.ndarray:
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    _check_host(lib().pth_tonemap_rgb8(_ptr(rgba), w, h, _ptr(out)), "pth_tonemap_rgb8")
    return out


def write_ppm(path: str, rgba: np.ndarray):
    """Tonemap like shaders/framebuffer.frag and write a binary PPM (top row first)."""
    rgb = tonemap_rgb8(rgba)
    _check_host(lib().pth_write_ppm(path.encode(), _ptr(rgb), rgb.shape[1], rgb.shape[0]), "pth_write_ppm")


def write_pfm(path: str, rgba: np.ndarray):
    a = np.ascontiguousarray(rgba, np.float32)
    _check_host(lib().pth_write_pfm(path.encode(), _ptr(a), a.shape[1], a.shape[0]), "pth_write_pfm")


def read_pfm(path: str, width: int, height: int) -> np.ndarray:
    out = np.zeros((height, width, 4), np.float32)
    _check_host(lib().pth_read_pfm(path.encode(), _ptr(out), width, height), "pth_read_pfm")
    return out
