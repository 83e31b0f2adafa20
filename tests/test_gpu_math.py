"""-m gpu: the device scalar routines (jello_amd/csrc/dmath.h) must agree bit for bit with the
oracle's (oracle/omath.h) -- this is what makes flatten's line counts reproducible -- and the
build must not contract a*b+c into an FMA."""
import ctypes

import numpy as np
import pytest

from oracle import oracle_engine

pytestmark = pytest.mark.gpu
FP = ctypes.POINTER(ctypes.c_float)


def gpu_op(engine, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float32)
        bp = b.ctypes.data_as(FP)
    engine.hip.jh_selftest_math.argtypes = [ctypes.c_void_p, ctypes.c_int, FP, FP, FP, ctypes.c_uint32]
    rc = engine.hip.jh_selftest_math(engine.ctx, op, a.ctypes.data_as(FP), bp, out.ctypes.data_as(FP), a.size)
    assert rc == 0
    return out


def cpu_vec(name, a, b=None):
    L = oracle_engine.lib()
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    if b is None:
        getattr(L, name)(a.ctypes.data_as(FP), out.ctypes.data_as(FP), ctypes.c_int(a.size))
    else:
        b = np.ascontiguousarray(b, dtype=np.float32)
        getattr(L, name)(a.ctypes.data_as(FP), b.ctypes.data_as(FP), out.ctypes.data_as(FP), ctypes.c_int(a.size))
    return out


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


N = 1 << 21


@pytest.mark.parametrize("op,name,lo,hi", [(0, "oracle_vec_sin", -40, 40), (1, "oracle_vec_cos", -40, 40), (3, "oracle_vec_acos", -1, 1),
                                           (4, "oracle_vec_asin", -1, 1), (5, "oracle_vec_pow23", -8, 8)])
def test_unary_transcendentals_bit_exact(engine, op, name, lo, hi):
    rng = np.random.default_rng(op + 1)
    x = (rng.random(N) * (hi - lo) + lo).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 0.5, 1e-30, -1e-30, 0.78539816]
    assert np.array_equal(bits(gpu_op(engine, op, x)), bits(cpu_vec(name, x)))


def test_atan2_bit_exact(engine):
    rng = np.random.default_rng(7)
    y = (rng.standard_normal(N) * 10 ** rng.uniform(-6, 6, N)).astype(np.float32)
    x = (rng.standard_normal(N) * 10 ** rng.uniform(-6, 6, N)).astype(np.float32)
    y[:6] = [0.0, -0.0, 0.0, 1.0, -1.0, 0.0]
    x[:6] = [0.0, -1.0, -1.0, 0.0, 0.0, 1.0]
    assert np.array_equal(bits(gpu_op(engine, 2, y, x)), bits(cpu_vec("oracle_vec_atan2", y, x)))


def test_ieee_div_sqrt_round_and_no_fma(engine):
    rng = np.random.default_rng(11)
    a = (rng.standard_normal(N) * 10 ** rng.uniform(-20, 20, N)).astype(np.float32)
    b = (rng.standard_normal(N) * 10 ** rng.uniform(-20, 20, N)).astype(np.float32)
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(gpu_op(engine, 6, a, b)), bits(a / b)), "f32 division is not correctly rounded"
        assert np.array_equal(bits(gpu_op(engine, 7, np.abs(a))), bits(np.sqrt(np.abs(a)))), "f32 sqrt is not correctly rounded"
        c = (rng.random(N) * 4 - 2).astype(np.float32)
        d = (rng.random(N) * 4 - 2).astype(np.float32)
        assert np.array_equal(bits(gpu_op(engine, 12, c, d)), bits((c * d).astype(np.float32) + c)), "a*b+a was contracted to an FMA"
    h = (np.arange(-2000, 2000, dtype=np.float32) * 0.5)
    assert np.array_equal(gpu_op(engine, 8, h), np.rint(h)), "round() must be ties-to-even"
    sat = np.array([-1.0, 0.0, 0.9, 1.0, 4.2949673e9, 5e9, np.nan, -np.inf, np.inf, 2.5], dtype=np.float32)
    assert list(gpu_op(engine, 9, sat).view(np.uint32)) == [0, 0, 0, 1, 0xffffffff, 0xffffffff, 0, 0, 0xffffffff, 2]
    assert list(gpu_op(engine, 10, sat).view(np.int32)) == [-1, 0, 0, 1, 2147483647, 2147483647, 0, -2147483648, 2147483647, 2]
    # (round 6: the conversions are single hardware instructions now -- every edge of their range against the definition)
    edge = np.array([-2147483648.0, -2147483904.0, -2147483520.0, 2147483520.0, 2147483648.0, 4294967040.0, 4294967296.0, -3e9, -0.9, -1e-40, 1e-40,
                     -0.0, 16777217.0, -16777216.0, 3.9999998, -3.9999998, 1e38, -1e38], dtype=np.float32)

    def ref_u(f):
        return 0 if not f > 0 else (0xffffffff if f >= 4294967296.0 else int(f))

    def ref_i(f):
        return 0 if f != f else (2147483647 if f >= 2147483648.0 else (-2147483648 if f <= -2147483648.0 else int(f)))
    assert list(gpu_op(engine, 9, edge).view(np.uint32)) == [ref_u(float(f)) for f in edge]
    assert list(gpu_op(engine, 10, edge).view(np.int32)) == [ref_i(float(f)) for f in edge]
    rng2 = np.random.default_rng(12)
    wide = (rng2.standard_normal(1 << 18) * 10 ** rng2.uniform(-3, 11, 1 << 18)).astype(np.float32)
    assert np.array_equal(gpu_op(engine, 9, wide).view(np.uint32), np.array([ref_u(float(f)) for f in wide], dtype=np.uint64).astype(np.uint32))
    assert np.array_equal(gpu_op(engine, 10, wide).view(np.int32), np.array([ref_i(float(f)) for f in wide], dtype=np.int64).astype(np.int32))


def test_f16_store_conversion(engine):
    rng = np.random.default_rng(5)
    x = np.concatenate([(rng.standard_normal(1 << 18) * 10 ** rng.uniform(-9, 5, 1 << 18)).astype(np.float32),
                        np.array([0, 65504, 65519.99, 65520, 1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5], dtype=np.float32)])
    got = gpu_op(engine, 11, x).view(np.uint32).astype(np.uint16)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)


def test_min_max_clamp_semantics(engine):
    """min/max are IEEE minNum/maxNum with -0 < +0 on both sides (oracle/omath.h <-> v_min_f32/v_max_f32)."""
    rng = np.random.default_rng(13)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 0.5, 1e-40, -1e-40, 2.0, -3.0], dtype=np.float32)
    a, b = np.meshgrid(special, special)
    a = np.concatenate([a.ravel(), rng.standard_normal(1 << 16).astype(np.float32)])
    b = np.concatenate([b.ravel(), rng.standard_normal(1 << 16).astype(np.float32)])
    L = oracle_engine.lib()
    mn, mx, cl = np.empty_like(a), np.empty_like(a), np.empty_like(a)
    L.oracle_vec_minmaxclamp(a.ctypes.data_as(FP), b.ctypes.data_as(FP), mn.ctypes.data_as(FP), mx.ctypes.data_as(FP), cl.ctypes.data_as(FP), ctypes.c_int(a.size))

    def canon(x):
        x = bits(x).copy()
        x[(x & 0x7fffffff) > 0x7f800000] = 0x7fc00000
        return x
    assert np.array_equal(canon(gpu_op(engine, 14, a, b)), canon(mn))
    assert np.array_equal(canon(gpu_op(engine, 15, a, b)), canon(mx))
    assert np.array_equal(canon(gpu_op(engine, 16, a)), canon(cl))
    with np.errstate(all="ignore"):
        prod = (a * b).astype(np.float32)
    cl2 = np.empty_like(a)
    L.oracle_vec_minmaxclamp(prod.ctypes.data_as(FP), b.ctypes.data_as(FP), mn.ctypes.data_as(FP), mx.ctypes.data_as(FP), cl2.ctypes.data_as(FP), ctypes.c_int(a.size))
    assert np.array_equal(canon(gpu_op(engine, 17, a, b)), canon(cl2))


@pytest.mark.parametrize("form", [0, 1, 2])
def test_divergent_value_atomics_behave_like_a_serial_execution(engine, form):
    """jh_selftest_atomics: per-lane-value returning atomics on one address inside a loop that lanes skip and leave at
    different trips (the pattern LLVM's atomic optimizer miscompiled in its DPP strategy), plain (0), through
    wave_bump (1) and the LDS forms (2): every range handed out tiles [0, counter) exactly / the words equal the serial result."""
    engine.hip.jh_selftest_atomics.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    engine.hip.jh_selftest_atomics.restype = ctypes.c_int
    for seed, waves in ((1, 1), (2, 4), (3, 37), (4, 1024), (5, 4096)):
        assert engine.hip.jh_selftest_atomics(engine.ctx, form, seed, waves) == 0, (form, seed, waves)
    assert engine.hip.jh_selftest_atomics(engine.ctx, 3, 1, 1) < 0
    assert engine.hip.jh_selftest_atomics(engine.ctx, 0, 1, 0) < 0
