"""CPU: the oracle's binary64 transcendental kernels (oracle/omath.h) round to the same binary32
values as libm on millions of samples, round() is ties-to-even, u32()/i32() saturate."""
import ctypes

import numpy as np
import pytest

from oracle import oracle_engine

FP = ctypes.POINTER(ctypes.c_float)


def vec(name, a, b=None):
    L = oracle_engine.lib()
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    if b is None:
        getattr(L, name)(a.ctypes.data_as(FP), out.ctypes.data_as(FP), ctypes.c_int(a.size))
    else:
        b = np.ascontiguousarray(b, dtype=np.float32)
        getattr(L, name)(a.ctypes.data_as(FP), b.ctypes.data_as(FP), out.ctypes.data_as(FP), ctypes.c_int(a.size))
    return out


N = 1_000_000


@pytest.mark.parametrize("name,fn,lo,hi", [("oracle_vec_sin", np.sin, -50, 50), ("oracle_vec_cos", np.cos, -50, 50),
                                           ("oracle_vec_acos", np.arccos, -1, 1), ("oracle_vec_asin", np.arcsin, -1, 1)])
def test_matches_libm_rounded_once(built, name, fn, lo, hi):
    x = (np.random.default_rng(1).random(N) * (hi - lo) + lo).astype(np.float32)
    got = vec(name, x)
    want = fn(x.astype(np.float64)).astype(np.float32)
    assert np.count_nonzero(got != want) == 0


def test_atan2_and_pow23(built):
    rng = np.random.default_rng(2)
    y = (rng.standard_normal(N) * 10 ** rng.uniform(-6, 6, N)).astype(np.float32)
    x = (rng.standard_normal(N) * 10 ** rng.uniform(-6, 6, N)).astype(np.float32)
    assert np.count_nonzero(vec("oracle_vec_atan2", y, x) != np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)) == 0
    z = (rng.random(N) * 16 - 8).astype(np.float32)
    want = (np.cbrt(np.abs(z.astype(np.float64))) ** 2).astype(np.float32)
    got = vec("oracle_vec_pow23", z)
    assert np.max(np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))) <= 1
    assert np.count_nonzero(got != want) < N * 1e-5
    # special values follow libm's atan2 conventions
    L = oracle_engine.lib()
    L.oracle_atan2.restype = ctypes.c_float
    L.oracle_atan2.argtypes = [ctypes.c_float, ctypes.c_float]
    assert L.oracle_atan2(0.0, 0.0) == 0.0
    assert L.oracle_atan2(0.0, -1.0) == np.float32(np.pi)
    assert L.oracle_atan2(-0.0, -1.0) == -np.float32(np.pi)
    assert L.oracle_atan2(1.0, 0.0) == np.float32(np.pi / 2)


def test_round_is_ties_to_even_and_casts_saturate(built):
    L = oracle_engine.lib()
    L.oracle_round.restype = ctypes.c_float
    L.oracle_round.argtypes = [ctypes.c_float]
    L.oracle_to_u32.restype = ctypes.c_uint32
    L.oracle_to_u32.argtypes = [ctypes.c_float]
    L.oracle_to_i32.restype = ctypes.c_int32
    L.oracle_to_i32.argtypes = [ctypes.c_float]
    assert [L.oracle_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 3.5)] == [0.0, 2.0, 2.0, -0.0, -2.0, 2.0, 4.0]
    assert [L.oracle_to_u32(v) for v in (-1.0, 0.0, 0.99, 7.9, 5e9, float("nan"), float("inf"))] == [0, 0, 0, 7, 0xffffffff, 0, 0xffffffff]
    assert [L.oracle_to_i32(v) for v in (-1.9, 1.9, 3e9, -3e9, float("nan"))] == [-1, 1, 2147483647, -2147483648, 0]
