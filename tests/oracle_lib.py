"""ctypes view of oracle/liboracle.so (the CPU checker).  Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by pathtracer_amd."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_float, c_int, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libptref.so")

OP = dict(MIN=0, MAX=1, DIV=2, SQRT=3, F2H2F=4, SIN=5, COS=6, TAN=7, ACOS=8, ATAN2=9, HADD=10, HMUL=11, RAND=12, OREN_NAYAR=13, GGX_NEE=14, GGX_IS=15)
FLAG_LIBM = 1
FLAG_EST_NEE = 2
FLAG_EST_BRDF = 4


class oracle_scene(ctypes.Structure):
    _fields_ = [("spheres", c_void_p), ("n_spheres", c_uint32),
                ("tris", c_void_p), ("n_tris", c_uint32),
                ("light_tris", c_void_p), ("n_light_tris", c_uint32),
                ("materials", c_void_p), ("n_materials", c_uint32),
                ("nodes", c_void_p), ("n_nodes", c_uint32),
                ("light_spheres", c_void_p), ("n_light_spheres", c_uint32),
                ("env_rgb8", c_void_p), ("env_w", c_uint32), ("env_h", c_uint32),
                ("tex_rgb8", c_void_p), ("tex_layers", c_uint32),
                ("tex_wrap_s", c_uint32), ("tex_wrap_t", c_uint32), ("tex_mag_filter", c_uint32)]


class oracle_counters(ctypes.Structure):
    _fields_ = [("paths", c_uint64), ("rays_primary", c_uint64), ("rays_shadow", c_uint64), ("rays_bounce", c_uint64),
                ("node_steps", c_uint64), ("tri_tests", c_uint64), ("max_stack", c_uint64), ("reserved", c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        L = ctypes.CDLL(ORACLE_SO)
        L.oracle_f2h.restype = c_uint16
        L.oracle_f2h.argtypes = [c_float]
        L.oracle_h2f.restype = c_float
        L.oracle_h2f.argtypes = [c_uint16]
        L.oracle_math.argtypes = [c_int, c_void_p, c_void_p, c_uint32, c_void_p]
        L.oracle_trace_closest.argtypes = [POINTER(oracle_scene), c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]
        L.oracle_render.argtypes = [POINTER(oracle_scene), c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_void_p, c_uint32,
                                    c_void_p, c_uint32, c_uint32, POINTER(oracle_counters)]
        L.oracle_pcg_stream.argtypes = [c_uint32, c_uint32, c_uint32, c_uint32, c_void_p]
        L.oracle_probe.argtypes = [c_int, c_void_p, c_uint32, c_void_p, c_uint32]
        L.oracle_estimate.argtypes = [POINTER(oracle_scene), c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def scene_from_desc(d, env_rgb8=None, tex_rgb8=None, tex_sampler=None):
    """Build an oracle_scene from a pathtracer_amd pt_scene_desc (same field order by design)."""
    s = oracle_scene()
    for name, _ in oracle_scene._fields_:
        setattr(s, name, getattr(d, name))
    keep = []
    if env_rgb8 is not None:
        env = np.ascontiguousarray(env_rgb8, np.uint8)
        s.env_rgb8, s.env_h, s.env_w = env.ctypes.data, env.shape[0], env.shape[1]
        keep.append(env)
    if tex_rgb8 is not None:
        tex = np.ascontiguousarray(tex_rgb8, np.uint8)
        s.tex_rgb8, s.tex_layers = tex.ctypes.data, tex.shape[0]
        keep.append(tex)
    if tex_sampler is not None:
        s.tex_wrap_s, s.tex_wrap_t, s.tex_mag_filter = tex_sampler
    s._keep = keep
    return s


def math(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a)
    rc = load().oracle_math(OP[op] if isinstance(op, str) else op, _p(a), _p(bb), a.size, _p(out))
    assert rc == 0
    return out


def trace_closest(osc, ro, rd, counts=False):
    ro = np.ascontiguousarray(ro, np.float32).reshape(-1, 3)
    rd = np.ascontiguousarray(rd, np.float32).reshape(-1, 3)
    n = ro.shape[0]
    prim = np.zeros(n, np.uint32)
    t_h = np.zeros(n, np.uint16)
    kind = np.zeros(n, np.uint8)
    uv = np.zeros((n, 2), np.float32)
    steps = np.zeros(n, np.uint32) if counts else None
    tests = np.zeros(n, np.uint32) if counts else None
    rc = load().oracle_trace_closest(byref(osc), _p(ro), _p(rd), n, _p(prim), _p(t_h), _p(kind), _p(uv), _p(steps), _p(tests))
    assert rc == 0
    if counts:
        return prim, t_h, kind, uv, steps, tests
    return prim, t_h, kind, uv


def render(osc, cam12, width, height, seeds, first_frame=0, bounces=5, flags=0, threads=0, rgba=None):
    cam = np.ascontiguousarray(cam12, np.float32).reshape(12)
    seeds = np.ascontiguousarray(seeds, np.uint32)
    if rgba is None:
        rgba = np.zeros((height, width, 4), np.float32)
    cnt = oracle_counters()
    rc = load().oracle_render(byref(osc), _p(cam), width, height, first_frame, seeds.size, _p(seeds), bounces, _p(rgba), flags,
                              threads, byref(cnt))
    assert rc == 0
    return rgba, cnt.as_dict()


# oracle_probe: name -> (code, floats in, floats out) per record (oracle/oracle.h)
PROBE = dict(MAP_TO_UNIT_SPHERE=(0, 2, 3), TNB_ROWS=(1, 3, 9), MAP_TO_TRIANGLE=(2, 11, 3), COSINE_HEMISPHERE=(3, 5, 3), NDF=(4, 4, 1),
             PICK_WI=(5, 6, 8), CALC_BRDF=(6, 18, 3), OCT_DECODE=(7, 1, 3), INVERSE3=(8, 9, 9), PRIMARY_RAY=(9, 17, 7))


def probe(what, records, flags=0):
    code, n_in, n_out = PROBE[what]
    a = np.ascontiguousarray(records, np.float32).reshape(-1, n_in)
    out = np.zeros((a.shape[0], n_out), np.float32)
    rc = load().oracle_probe(code, _p(a), a.shape[0], _p(out), flags)
    assert rc == 0
    return out


def estimate(osc, ro, rd, n_samples, bounces=5, flags=0, seed=1):
    """Mean radiance (rgb, float64) of n_samples estimator evaluations for one fixed ray."""
    ro = np.ascontiguousarray(ro, np.float32).reshape(3)
    rd = np.ascontiguousarray(rd, np.float32).reshape(3)
    out = np.zeros(3, np.float64)
    rc = load().oracle_estimate(byref(osc), _p(ro), _p(rd), n_samples, bounces, flags, seed, _p(out))
    assert rc == 0
    return out


def pcg_stream(x, y, seed, n):
    out = np.zeros(n, np.uint32)
    load().oracle_pcg_stream(x, y, seed, n, _p(out))
    return out


def load_ref():
    """oracle/_ref/libptref.so: the reference's own host TUs.  Only exists in the build container."""
    if not os.path.exists(REF_SO):
        if not os.path.isdir("/root/reference"):
            return None
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    R = ctypes.CDLL(REF_SO)
    R.ref_pcg32_next.restype = c_uint32
    R.ref_calculate_bvh.restype = c_uint32
    R.ref_find_emissive_tris.restype = c_uint32
    R.ref_find_emissive_spheres.restype = c_uint32
    return R
