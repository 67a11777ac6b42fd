"""The ABI's deterministic scalar math (include/dhmc_detmath.h) against libm.

These functions replace Julia's exp/log/log1p/randn/randexp on the hot path (call sites in the
header).  Bar: within 2 ulp of numpy/libm over the ranges the sampler uses, so the substitution
stays inside the reference's own tolerance class.  (ABI v2, round 4: table-driven; the tables are
what tools/gen_detmath_tables.py produces from 200-bit arithmetic.)
"""
import os
import subprocess
import sys

import numpy as np
import oracle_lib as ol

RNG = np.random.default_rng(0x23EF614D)


def ulp_err(got, ref):
    ref = np.asarray(ref, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_exp():
    x = np.concatenate([RNG.uniform(-745, 709, 200000), RNG.uniform(-40, 5, 200000), RNG.normal(0, 1e-3, 1000)])
    assert ulp_err(ol.detmath(0, x), np.exp(x)).max() <= 2.0
    assert ol.detmath(0, [-np.inf])[0] == 0.0
    assert ol.detmath(0, [np.inf])[0] == np.inf
    assert ol.detmath(0, [0.0])[0] == 1.0
    assert np.isnan(ol.detmath(0, [np.nan])[0])


def test_log():
    x = np.concatenate([np.exp(RNG.uniform(-700, 700, 200000)), RNG.uniform(0.5, 2.0, 200000),
                        RNG.uniform(0, 1, 100000), [5e-324, 1e-310, 10.0, 1.0]])
    assert ulp_err(ol.detmath(1, x), np.log(x)).max() <= 2.0
    assert ol.detmath(1, [1.0])[0] == 0.0
    assert ol.detmath(1, [0.0])[0] == -np.inf
    assert np.isnan(ol.detmath(1, [-1.0])[0])
    assert ol.detmath(1, [np.inf])[0] == np.inf


def test_log1p_and_logaddexp():
    u = np.concatenate([RNG.uniform(0, 1, 100000), np.exp(-RNG.uniform(0, 745, 100000))])
    assert ulp_err(ol.detmath(2, u), np.log1p(u)).max() <= 2.0
    x = RNG.uniform(-50, 10, 100000); y = x + RNG.normal(0, 20, 100000)
    got = ol.detmath(8, x, y)
    # max + log1p(.) cancels when the result is near 0: the bound is absolute, in ulps of max(x, y)
    assert (np.abs(got - np.logaddexp(x, y)) / np.spacing(np.maximum(np.abs(np.maximum(x, y)), 1.0))).max() <= 2.0
    # the -Inf rules the tree engine relies on (trees.jl:145 with ω = -Inf; NUTS.jl:79)
    ninf = -np.inf
    assert ol.detmath(8, [ninf], [ninf])[0] == ninf
    assert ol.detmath(8, [ninf], [-3.5])[0] == -3.5
    assert ol.detmath(8, [-3.5], [ninf])[0] == -3.5
    # symmetric bit for bit
    assert np.array_equal(got, ol.detmath(8, y, x))


def test_logistic_link_is_exp_division_and_log1p():
    """det_logistic_sigma / det_log1pexp (the logistic family's link; what oracle/targets.hpp LogisticTarget computes observation by
    observation, and what the kernels must reproduce from one shared reciprocal) are t = exp(-|η|), ONE IEEE division and log1p(t)."""
    eta = np.concatenate([RNG.normal(0, 3, 100000), RNG.uniform(-760, 760, 100000), [0.0, -0.0, 36.8, -36.8, 707.0, -707.0, np.inf, -np.inf]])
    t = ol.detmath(0, -np.abs(eta))
    with np.errstate(over="ignore"):
        sig = np.where(eta >= 0, 1.0 / (1.0 + t), t / (1.0 + t))
        l1pe = np.where(eta > 0, eta, 0.0) + ol.detmath(2, t)
    assert np.array_equal(ol.detmath(10, eta), sig)
    assert np.array_equal(ol.detmath(11, eta), l1pe)
    assert np.isnan(ol.detmath(10, [np.nan])[0]) and np.isnan(ol.detmath(11, [np.nan])[0])
    fin = np.isfinite(eta)
    from scipy.special import expit
    mid = np.abs(eta) < 700                      # (beyond: subnormal results, where an ulp is not a relative measure)
    assert ulp_err(sig[mid], expit(eta[mid])).max() <= 4.0      # (expit itself is not correctly rounded)
    assert np.abs(l1pe[fin] - np.logaddexp(0.0, eta[fin])).max() <= 2 * np.spacing(760.0)


def test_sincos2pi():
    a = np.concatenate([RNG.uniform(0, 1, 300000), [0.0, 0.25, 0.5, 0.75, 0.125, 1 - 2.0**-53]])
    s, c = ol.detmath(3, a), ol.detmath(4, a)
    # reference in x87 extended precision with an exact quadrant reduction (sin(2*pi*a) computed
    # naively in double loses relative accuracy near the zeros through the rounding of 2*pi)
    al = a.astype(np.longdouble)
    pi_ld = np.arctan(np.longdouble(1)) * 4
    t = 4 * al
    n = np.floor(t + np.longdouble(0.5))
    x = (t - n) * pi_ld / 2
    s0, c0 = np.sin(x), np.cos(x)
    q = n.astype(np.int64) & 3
    sr = np.choose(q, [s0, c0, -s0, -c0]).astype(np.float64)
    cr = np.choose(q, [c0, -s0, -c0, s0]).astype(np.float64)
    m = sr != 0
    assert ulp_err(s[m], sr[m]).max() <= 2.0
    m = cr != 0
    assert ulp_err(c[m], cr[m]).max() <= 2.0
    assert np.abs(s * s + c * c - 1).max() < 5e-16
    assert ol.detmath(3, [0.25])[0] == 1.0 and ol.detmath(4, [0.5])[0] == -1.0


def _bits_as_double(u64):
    return np.asarray(u64, np.uint64).view(np.float64)


def test_randexp_randn_moments():
    n = 400000
    r1 = RNG.integers(0, 2**64, n, dtype=np.uint64)
    r2 = RNG.integers(0, 2**64, n, dtype=np.uint64)
    e = ol.detmath(5, _bits_as_double(r1))
    assert np.all(e >= 0) and np.isfinite(e).all()
    assert abs(e.mean() - 1) < 0.01 and abs(e.var() - 1) < 0.02
    z0 = ol.detmath(6, _bits_as_double(r1), _bits_as_double(r2))
    z1 = ol.detmath(7, _bits_as_double(r1), _bits_as_double(r2))
    for z in (z0, z1):
        assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.01
        assert abs((z**4).mean() - 3) < 0.06
    assert abs(np.corrcoef(z0, z1)[0, 1]) < 0.01
    # extreme bit patterns stay finite
    ext = np.array([0, 2**64 - 1, 1 << 11, (1 << 11) - 1], np.uint64)
    assert np.isfinite(ol.detmath(5, _bits_as_double(ext))).all()
    assert np.isfinite(ol.detmath(6, _bits_as_double(ext), _bits_as_double(ext[::-1].copy()))).all()


def test_pow():
    m = np.arange(2, 5000, dtype=np.float64)
    got = ol.detmath(9, m, np.full_like(m, -0.75))
    assert (np.abs(got - m**-0.75) / m**-0.75).max() < 4e-15


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [hex(v) for v in ol.philox([0, 0, 0, 0], [0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in ol.philox([0xffffffff] * 4, [0xffffffff] * 2)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in ol.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_wave_dot_matches_plain_dot():
    for n in (1, 3, 30, 64, 65, 100, 1000, 1024):
        a = RNG.normal(size=n); b = RNG.normal(size=n)
        assert abs(ol.wave_dot(a, b) - float(np.dot(a, b))) <= 1e-13 * np.abs(a * b).sum()
    assert ol.wave_dot(np.array([np.nan]), np.array([1.0])) != ol.wave_dot(np.array([np.nan]), np.array([1.0]))  # NaN


def test_tables_are_what_the_generator_produces():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_detmath_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_softplus_table_against_high_precision():
    import mpmath as mp
    mp.mp.prec = 120
    d = np.concatenate([RNG.uniform(0, 16, 4000), RNG.uniform(16, 745, 1000), (np.arange(257) / 16.0)[:-1], np.nextafter(np.arange(1, 257) / 16.0, 0)])
    got = ol.detmath(8, np.zeros_like(d), -d)          # logaddexp(0, -d) = softplus(-d)
    ref = np.array([float(mp.log1p(mp.exp(-mp.mpf(float(v))))) for v in d])
    assert ulp_err(got, ref).max() <= 2.0


def test_exp_log_special_paths():
    # results in the subnormal range take the two-multiplication path; 709.78... is the last finite argument
    x = np.array([-708.4, -710.0, -730.0, -744.9, -745.13, 709.782712893383])
    ref = np.exp(x)
    got = ol.detmath(0, x)
    assert (np.abs(got - ref) <= np.maximum(np.spacing(ref), 5e-324)).all()
    assert np.isfinite(got[-1]) and ol.detmath(0, [709.7827128933841])[0] == np.inf
    # every cell border of the log table, from both sides
    cells = 1 + (np.arange(129) - 0.5) / 128
    xs = np.concatenate([cells, np.nextafter(cells, 0), np.nextafter(cells, 4), cells / 2, cells * 4])
    assert ulp_err(ol.detmath(1, xs), np.log(xs)).max() <= 2.0
