"""Device primitives of the numerics contract vs the oracle's, bit for bit (through pt_debug_eval)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pt):
    c = pt.PathTracer(8, 8)
    yield c
    c.close()


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def _specials():
    return np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, 65504.0, 65520.0, 65519.9,
                     5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, 0.001, 100.0, 0.5, 2.0], np.float32)


def _same(a, b):
    a, b = _bits(a), _bits(b)
    nan_a = (a & 0x7fffffff) > 0x7f800000
    nan_b = (b & 0x7fffffff) > 0x7f800000
    return np.array_equal(nan_a, nan_b) and np.array_equal(a[~nan_a], b[~nan_b])


@pytest.mark.parametrize("op", ["MIN", "MAX", "DIV", "HADD", "HMUL", "ATAN2"])
def test_binary_ops(ctx, ol, op):
    rng = np.random.default_rng(3)
    sp = _specials()
    a = np.concatenate([np.repeat(sp, sp.size), (rng.normal(size=200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)])
    b = np.concatenate([np.tile(sp, sp.size), (rng.normal(size=200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)])
    if op in ("HADD", "HMUL"):
        with np.errstate(over="ignore", invalid="ignore"):
            a = a.astype(np.float16).astype(np.float32)
            b = b.astype(np.float16).astype(np.float32)
    assert _same(ctx.debug_eval(ol.OP[op], a, b), ol.math(op, a, b))


@pytest.mark.parametrize("op,lo,hi", [("SQRT", 0.0, 1e6), ("F2H2F", -70000.0, 70000.0), ("SIN", -8.0, 8.0), ("COS", -8.0, 8.0),
                                      ("TAN", -1.6, 3.2), ("ACOS", -1.01, 1.01)])
def test_unary_ops(ctx, ol, op, lo, hi):
    rng = np.random.default_rng(4)
    a = np.concatenate([_specials(), rng.uniform(lo, hi, 400000).astype(np.float32),
                        (rng.normal(size=100000) * np.exp(rng.uniform(-30, 10, 100000))).astype(np.float32)])
    if op in ("SIN", "COS", "TAN"):
        # the contract's quadrant reduction is specified for |x| < 2^20 (the shader never leaves [-2pi, 2pi]);
        # beyond int range the float->int conversion itself is implementation-defined
        a = a[~(np.abs(a) >= 2.0 ** 20)]
    assert _same(ctx.debug_eval(ol.OP[op], a), ol.math(op, a))


def test_rcp_of_fp16_values(ctx, ol):
    """k_trace computes 1/rd for directions that went through fp16 with a 5-instruction reciprocal: it must be the IEEE
    quotient for EVERY fp16 value (zeros, denormals, infinities, NaNs included) -- compared with the device's own
    correctly rounded division (which test_binary_ops pins to the oracle's) and with the oracle directly."""
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    got = ctx.debug_eval(100, h)
    assert _same(got, ctx.debug_eval(ol.OP["DIV"], np.ones_like(h), h))
    assert _same(got, ol.math("DIV", np.ones_like(h), h))


def test_rand_clamp(ctx, ol):
    st = np.random.default_rng(5).integers(0, 2**32, 300000, dtype=np.uint64).astype(np.uint32)
    st[:4] = [0, 1, 0xffffffff, 0x80000000]
    assert _same(ctx.debug_eval(ol.OP["RAND"], st.view(np.float32)), ol.math("RAND", st.view(np.float32)))


@pytest.mark.parametrize("op", ["OREN_NAYAR", "GGX_NEE", "GGX_IS"])
def test_brdf_probes(ctx, ol, op):
    rng = np.random.default_rng(6)
    st = rng.integers(0, 2**32, 400000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    rough = rng.choice(np.array([0.01, 0.1, 0.15, 0.175, 0.2, 0.5, 1.0, 2.0], np.float32), st.size)
    g, o = ctx.debug_eval(ol.OP[op], st, rough), ol.math(op, st, rough)
    bad = np.nonzero(g.view(np.uint32) != o.view(np.uint32))[0]
    for i in bad[:10]:
        print(op, "state %08x rough %r gpu %r cpu %r" % (st.view(np.uint32)[i], rough[i], g[i], o[i]))
    assert _same(g, o)
