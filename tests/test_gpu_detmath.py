"""GPU: the ABI's scalar math on the device equals the CPU side of the same header, bit for bit, in every operand
placement the kernels use (csrc/detmath_dev.hpp: compiled as is, per-lane, wave-uniform with scalar loads and asm Horner
chains).  The whole-run parity tests depend on this; here it is checked function by function over wide input ranges, so that
a mismatch names the function instead of showing up as a diverged chain.  Reference call sites of the functions:
trees.jl:145, NUTS.jl:44,70,87, stepsize.jl:136-170, hamiltonian.jl:124 (include/dhmc_detmath.h)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(0xD37A)
POLICIES = {0: "generic", 1: "vector", 2: "uniform"}


def _dev(pkg, kind, policy, x, y=None):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, np.float64)
        yp = y.ctypes.data_as(C.c_void_p)
    rc = pkg.abi.lib().dhmc_detmath_selftest(0, kind, policy, C.c_int64(x.size), x.ctypes.data_as(C.c_void_p), yp,
                                             out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return out


def _bits(n):
    return RNG.integers(0, 2**64, n, dtype=np.uint64).view(np.float64)


def _inputs(kind, n):
    inf, nan = np.inf, np.nan
    if kind == 0:      # exp: whole range, the subnormal results, thresholds, specials
        x = np.concatenate([RNG.uniform(-746, 710, n), RNG.uniform(-40, 5, n), RNG.normal(0, 1e-3, n // 8),
                            RNG.uniform(-745.2, -708, n // 4),
                            [0.0, -0.0, 709.782712893384, 709.7827128933841, -745.2, -745.13, -708.3964, inf, -inf, nan, 1e-320]])
        return x, None
    if kind == 1:      # log: all binades, around 1, cell borders, subnormals, specials
        cells = 1 + (np.arange(129) - 0.5) / 128
        x = np.concatenate([np.exp(RNG.uniform(-744, 709, n)), RNG.uniform(0.5, 2.0, n), 1 + RNG.normal(0, 1e-6, n // 8),
                            cells, np.nextafter(cells, 0), np.nextafter(cells, 4), _bits(n // 4),
                            [5e-324, 1e-310, 2.2250738585072014e-308, 1.0, 0.0, -0.0, -1.0, inf, -inf, nan, 10.0]])
        return x, None
    if kind == 2:
        return np.concatenate([RNG.uniform(0, 1, n), np.exp(-RNG.uniform(0, 745, n)), [0.0, 1.0, 1e-17, inf]]), None
    if kind in (3, 4):
        edges = (np.arange(65) - 0.5) / 64
        return np.concatenate([RNG.uniform(0, 1, n), np.arange(64) / 64.0, np.clip(edges, 0, None), np.nextafter(np.clip(edges, 0, None), 1),
                               [1 - 2.0**-53, 2.0**-53, 0.0]]), None
    if kind == 5:
        return np.concatenate([_bits(n), np.array([0, 2**64 - 1, 1 << 11, (1 << 11) - 1], np.uint64).view(np.float64)]), None
    if kind in (6, 7):
        ext = np.array([0, 2**64 - 1, 1 << 11, (1 << 11) - 1], np.uint64).view(np.float64)
        return np.concatenate([_bits(n), ext]), np.concatenate([_bits(n), ext[::-1]])
    if kind == 8:      # logaddexp: close, far, the softplus table's far end, equal, infinite, NaN
        x = np.concatenate([RNG.uniform(-50, 10, n), RNG.uniform(-5, 5, n), np.zeros(n // 4), RNG.uniform(-800, 800, n // 4),
                            [0.0, -inf, -inf, -3.5, inf, inf, nan, 1.0, 0.0, 0.0, 0.0, -inf, 5.0]])
        y = np.concatenate([x[:n] + RNG.normal(0, 20, n), x[n:2 * n] + RNG.normal(0, 0.5, n), -RNG.uniform(15.9, 16.1, n // 4),
                            RNG.uniform(-800, 800, n // 4),
                            [0.0, -inf, -3.5, -inf, inf, 1.0, 1.0, nan, -16.0, -15.999999999999998, -745.0, inf, 5.0]])
        return x, y
    if kind == 9:
        m = np.arange(2, 2 + n, dtype=np.float64)
        return m, np.full_like(m, -0.75)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", range(10))
def test_device_scalar_math_equals_the_cpu_side_bit_for_bit(kind):
    pkg = load_package()
    n = 20000 if kind != 9 else 5000
    x, y = _inputs(kind, n)
    want = ol.detmath(kind, x, y)
    for policy, name in POLICIES.items():
        # the uniform policy runs one value per wavefront: a slice is enough, the edge cases sit at the end
        sl = slice(None) if policy != 2 else slice(-4000, None)
        got = _dev(pkg, kind, policy, x[sl], None if y is None else y[sl])
        w = want[sl]
        same = (got.view(np.uint64) == w.view(np.uint64)) | (np.isnan(got) & np.isnan(w))
        bad = np.flatnonzero(~same)
        assert bad.size == 0, (f"kind {kind}, policy {name}: {bad.size} of {w.size} differ, first x={x[sl][bad[0]]!r}"
                               f" y={None if y is None else y[sl][bad[0]]!r} device={got[bad[0]]!r} cpu={w[bad[0]]!r}")


def _link_inputs(n):
    """η for the logistic link: whole waves (128 consecutive values: two per lane) of ordinary arguments first — they take the batched common
    path, one reciprocal for both quotients —, among them |η| > 36.74 (1 + e^{-|η|} == 1), exact zeros (the padding observations: w == 2)
    and the last arguments before the rare path; then waves that mix in |η| > 707, infinities and NaN (whole-wave rare path)."""
    inf, nan = np.inf, np.nan
    n = (n // 128) * 128
    common = np.concatenate([RNG.normal(0, 3, n), RNG.uniform(-40, 40, n), RNG.uniform(-707, 707, n), RNG.normal(0, 1e-8, n),
                             np.repeat([0.0, -0.0, 707.0, -707.0, 36.7, -36.8, 1e-300, -1e-300], 16),
                             np.repeat([2.0**-52, -2.0**-52, 2.0**-53, 1.5 * 2.0**-53, -3e-16, 2.5e-16, 1e-15, -1e-15], 16),   # 1 + e^{-|η|} rounds to 2 or just below
                             RNG.uniform(-1e-15, 1e-15, n),
                             np.ldexp(RNG.uniform(0.5, 1, n), RNG.integers(-60, 10, n)) * RNG.choice([-1.0, 1.0], n)])
    assert common.size % 128 == 0 and np.all(np.abs(common) <= 707.0)
    rare = np.concatenate([RNG.uniform(-760, 760, n), RNG.normal(0, 3, 100),
                           [707.0000000000001, -707.0000000000001, 708.4, -708.4, 745.2, -745.2, 745.3, -746.0, 800.0, -800.0, inf, -inf, nan, 0.0, 5e-324]])
    return np.concatenate([common, rare])


@pytest.mark.parametrize("kind", [10, 11])
def test_logistic_link_equals_the_cpu_side_bit_for_bit(kind):
    """σ(η) and log(1 + e^η) of the logistic family as the round engine evaluates them (policy 1: logistic_link_batch) and as the
    wave-per-chain functor does (policy 0: the header's functions), against the CPU side's IEEE divisions."""
    pkg = load_package()
    x = _link_inputs(40960)
    want = ol.detmath(kind, x)
    for policy in (0, 1):
        got = _dev(pkg, kind, policy, x)
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        bad = np.flatnonzero(~same)
        assert bad.size == 0, (f"kind {kind}, policy {POLICIES[policy]}: {bad.size} of {want.size} differ, first η={x[bad[0]]!r}"
                               f" device={got[bad[0]]!r} cpu={want[bad[0]]!r}")


def test_selftest_rejects_bad_arguments():
    pkg = load_package()
    L = pkg.abi.lib()
    x = np.zeros(4)
    p = x.ctypes.data_as(C.c_void_p)
    assert L.dhmc_detmath_selftest(0, 99, 0, C.c_int64(4), p, None, p) == 1
    assert L.dhmc_detmath_selftest(0, 0, 7, C.c_int64(4), p, None, p) == 1
    assert L.dhmc_detmath_selftest(0, 8, 0, C.c_int64(4), p, None, p) == 1     # logaddexp needs y
    assert L.dhmc_detmath_selftest(0, 0, 0, C.c_int64(0), p, None, p) == 1
    assert L.dhmc_detmath_selftest(0, 10, 2, C.c_int64(4), p, None, p) == 1    # the link has no wave-uniform form
