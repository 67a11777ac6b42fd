"""Replay of test/test_NUTS.jl and the deterministic parts of test/test_stepsize.jl,
test/test_hamiltonian.jl against the oracle."""
import math

import numpy as np
import pytest
import oracle_lib as ol

RNG = np.random.default_rng(0x8332E05C)


def test_rand_bool_logprob():  # test_NUTS.jl:10-21
    for prob in np.arange(1, 10) / 10:
        cnt, used = ol.rand_bool_logprob(math.log(prob), 10000, seed=int(prob * 100))
        assert abs(cnt / 10000 - prob) <= 0.02
        assert used == 10000
    # logprob >= 0 never touches the RNG
    for lp in (0.0, 10.0):
        cnt, used = ol.rand_bool_logprob(lp, 10000)
        assert cnt == 10000 and used == 0


def test_low_level_turn_statistics():  # test_NUTS.jl:27-42
    p = np.ones(3); c = 0.1
    t1 = np.stack([p, p - c, p, p - c, p])
    t2 = np.stack([3 * p, 3 * p + c, 3 * p, 3 * p + c, 3 * p])
    t3 = np.stack([2 * p, 2 * p + c, 2 * p, 2 * p + c, -2 * p])
    turning, rho = ol.combine_turn(t1, t2)
    assert not turning
    assert np.array_equal(rho, t1[4] + t2[4])
    turning, _ = ol.combine_turn(t1, t3)
    assert turning


def test_low_level_visited_statistics():  # test_NUTS.jl:44-55
    for det in (True, False):
        assert np.isclose(ol.acceptance([math.log(0.3)], [0], det), 0.3)
        assert np.isclose(ol.acceptance([math.log(0.6)], [0], det), 0.6)
        a = ol.acceptance([math.log(0.3), math.log(0.3), math.log(0.6), math.log(10)], [0, 0, 0, 1], det)
        assert np.isclose(a, 0.4)


def test_unconditional_divergence():  # test_NUTS.jl:75-85
    o = ol.Oracle(3, 4, target=ol.TARGET_ALWAYS_DIVERGENT)
    o.init(np.zeros((4, 3)))
    o.set_stepsize(1.0)
    r = o.run(1)
    assert np.all(r["term_left"] == r["term_right"])      # is_divergent
    assert np.all(r["acceptance_rate"] == 0)
    assert np.all(r["depth"] == 0)
    assert np.all(r["steps"] == 1)
    assert np.all(r["draws"] == 0)                         # proposal unchanged


def test_logdensity_infinity_fallbacks():  # test_hamiltonian.jl:197-200
    assert ol.logdensity(-np.inf, [1.0], [1.0]) == -np.inf
    assert ol.logdensity(np.nan, [1.0], [1.0]) == -np.inf
    assert ol.logdensity(9.0, [np.nan], [1.0]) == -np.inf
    assert ol.logdensity(9.0, [2.0], [1.0]) == 7.0


def _leapfrog_gaussian(q, p, mu, prec, eps, m):  # test_hamiltonian.jl:72-78, independent formula
    u = np.sqrt(1 / m)
    grad = lambda x: -prec * (x - mu)
    ph = p + eps / 2 * grad(q)
    q1 = q + eps * u * (u * ph)
    p1 = ph + eps / 2 * grad(q1)
    return q1, p1


def test_leapfrog_calculation():  # test_hamiltonian.jl:69-109 (diagonal target variant)
    n = 3
    m = RNG.normal(size=n) ** 2 + 0.01
    mu = RNG.normal(size=n); prec = 1 / (RNG.normal(size=n) ** 2 + 0.01)
    q = RNG.normal(size=n); p = RNG.normal(size=n)
    eps = 0.05
    cfg = ol.make_config(n, 1, target=ol.TARGET_DIAG_NORMAL, params=np.concatenate([mu, prec]))
    qs, ps, _, _, st = ol.leapfrog(cfg, 1 / m, q, p, eps, 100)
    assert st == 0
    for i in range(100):
        q, p = _leapfrog_gaussian(q, p, mu, prec, eps, m)
        assert np.allclose(qs[i], q, rtol=math.sqrt(np.finfo(float).eps))
        assert np.allclose(ps[i], p, rtol=math.sqrt(np.finfo(float).eps))


def test_nan_position_flags_error():  # test_hamiltonian.jl:111-115 (throw -> status bit)
    o = ol.Oracle(3, 1)
    rc = o.init(np.full((1, 3), np.nan), allow_failure=True)
    assert rc == ol.ERR_CHAIN_FAILURE and o.status()[0] & ol.ST_NONFINITE_POSITION


def test_leapfrog_energy_and_reversibility():  # test_hamiltonian.jl:118-153
    for _ in range(50):
        n = 5
        minv = 1 / (RNG.normal(size=n) ** 2 + 0.01)
        mu = RNG.normal(size=n); prec = 1 / (RNG.normal(size=n) ** 2 + 0.01)
        q = mu + RNG.normal(size=n) / np.sqrt(prec); p = RNG.normal(size=n) / np.sqrt(minv)
        cfg = ol.make_config(n, 1, target=ol.TARGET_DIAG_NORMAL, params=np.concatenate([mu, prec]))
        eps = 0.1 * min(1.0, float(np.sqrt(1 / (prec * minv)).min()))
        qs, ps, pis, _, _ = ol.leapfrog(cfg, minv, q, p, eps, 10)
        pi0 = ol.leapfrog(cfg, minv, q, p, 0.0, 1)[2][0]
        assert np.all(np.abs(pis - pi0) < 0.5)            # :118-141
        q1, p1 = ol.leapfrog(cfg, minv, q, p, eps, 1)[:2]
        q2, p2 = ol.leapfrog(cfg, minv, q1[0], p1[0], -eps, 1)[:2]
        assert np.abs(p2[0] - p).max() <= 1e-5 and np.abs(q2[0] - q).max() <= 1e-6   # :143-153


def test_stepsize_general_rootfinding():  # test_stepsize.jl:9-25
    thr = math.log(0.8)
    rc, eps = ol.find_initial_stepsize_linear(-3.0)
    assert rc == 0 and eps == 0.05 and -3 * eps > thr > -3 * 0.1
    rc, eps = ol.find_initial_stepsize_linear(-3.0, initial_eps=0.01)
    assert rc == 0 and eps == 0.08 and -3 * eps < thr < -3 * 0.01
    rc, _ = ol.find_initial_stepsize_linear(0.0, intercept=1.0)   # constant A: the reference throws
    assert rc == 1
    assert ol.find_initial_stepsize_linear(-3.0, log_threshold=float("nan"))[0] == 2   # :13
    assert ol.find_initial_stepsize_linear(-3.0, log_threshold=1.0)[0] == 2            # :14
    assert ol.find_initial_stepsize_linear(-3.0, initial_eps=-0.5)[0] == 2             # :15
    assert ol.find_initial_stepsize_linear(-3.0, maxiter=2)[0] == 2                    # :16


def test_dual_averaging_known_values():  # test_stepsize.jl:42-44 + SURVEY.md §8a row a24
    for det in (True, False):
        st = ol.da_init(100.0, det)
        assert st[4] == 0 and st[1] == 1 and st[2] == 0
        st = ol.da_init(1.0, det)
        st = ol.da_adapt(st, 0.5, det=det)
        assert st[1] == 2 and np.isclose(st[2], 0.025, rtol=1e-14)
        assert np.isclose(st[3], 1.5954783118074982, rtol=1e-14)
        assert np.isclose(st[4], 0.9486770801170034, rtol=1e-14)
        st = ol.da_adapt(st, 1.0, det=det)
        assert st[1] == 3 and np.isclose(st[2], 0.0076923076923077, rtol=1e-12)
        assert np.isclose(st[3], 2.036115737983449, rtol=1e-14)
        assert np.isclose(st[4], 1.4257269995496586, rtol=1e-14)


def _dummy_acceptance_rate(eps, sigma, rng):  # test_stepsize.jl:33
    return min(1 / eps * math.exp(rng.normal() * sigma - sigma ** 2 / 2), 1)


@pytest.mark.parametrize("eps0,iters,sigma,atol", [(100.0, 500, 0.05, 0.02), (2.0, 2000, 0.05, 0.01),
                                                   (20.0, 10000, 2.0, 0.04)])
def test_dual_averaging_convergence(eps0, iters, sigma, atol):  # test_stepsize.jl:37-71
    rng = np.random.default_rng(1)
    delta = 0.65
    st = ol.da_init(eps0)
    for _ in range(iters):
        st = ol.da_adapt(st, _dummy_acceptance_rate(math.exp(st[3]), sigma, rng), delta=delta)
    fe = math.exp(st[4])
    mean_rate = np.mean([_dummy_acceptance_rate(fe, sigma, rng) for _ in range(10000)])
    assert abs(mean_rate - delta) <= atol


def test_gaussian_ke_full_and_diagonal():  # test_hamiltonian.jl:20-47
    for _ in range(10):
        K = int(RNG.integers(2, 11))
        A = RNG.normal(size=(K, K)); Sigma = A.T @ A + 0.01         # rand_Σ, test/utilities.jl:6-9
        Minv = np.linalg.inv(Sigma)
        ps, W = ol.rand_p_dense(Minv, 10000, seed=K)
        assert np.allclose(np.triu(W, 1), 0)                        # W isa LowerTriangular
        assert np.allclose(Minv @ W @ W.T, np.eye(K), atol=1e-7)    # M⁻¹ W Wᵀ ≈ I
        Cm = np.cov(ps.T)
        assert np.linalg.norm(Cm - Sigma) <= 0.1 * np.linalg.norm(Sigma)   # Matrix(Σ) ≈ C rtol = 0.1
    with pytest.raises(ValueError):
        ol.rand_p_dense(-np.eye(3), 1)
    # diagonal: W = Diagonal(sqrt.(1 ./ diag)), sample variance of rand_p ≈ 1 ./ diag(M⁻¹)
    K = 6
    d = RNG.normal(size=K) ** 2 + 0.01
    o = ol.Oracle(K, 1)
    o.set_metric_diag(1 / d)
    assert np.allclose(o.metric_diag()[0] * d, 1)


def test_logistic_gradient_in_blocks_of_observations():
    """include/dhmc.h: (Xᵀr)_d is summed block by block (DHMC_LOGISTIC_BLOCK observations, blocks added in ascending order).
    With more than two blocks the oracle's ℓ and ∇ℓ are still those of the model: ℓ from the probe, ∇ℓ recovered from one
    leapfrog step with p₀ = 0 and unit metric (q₁ - q₀ = ϵ²/2 ∇ℓ(q₀)), against numpy."""
    rng = np.random.default_rng(11)
    N, D = 5003, 5                                   # 3 blocks of 2048, the last one ragged
    X = rng.normal(size=(N, D)) / 4
    y = (rng.random(N) < 0.4).astype(float)
    q0 = rng.normal(size=D) / 3
    o = ol.Oracle(D, 1, target=ol.TARGET_LOGISTIC, params=ol.target_params_blob(ol.TARGET_LOGISTIC, D, X=X, y=y))
    o.init(q0[None, :])
    eps = 2.0 ** -6
    tr = o.leapfrog_trajectory(eps, 0, 1, p=np.zeros(D))
    eta = X @ q0
    lq = np.sum(y * eta - np.logaddexp(0.0, eta)) - 0.5 * q0 @ q0
    grad = X.T @ (y - 1 / (1 + np.exp(-eta))) - q0
    assert abs(tr["logdensity"][0, 0] - lq) < 1e-9 * abs(lq)
    got = (tr["q"][0, 1] - q0) * 2 / eps ** 2
    assert np.allclose(got, grad, rtol=0, atol=1e-7 * np.abs(grad).max())


def test_dense_one_product_recurrence_stays_next_to_the_references():
    """oracle/hamiltonian.hpp leapfrog: with `one_product` the dense leapfrog propagates p♯ = M⁻¹p and u = M⁻¹∇ℓq by
    linearity and takes ONE product per step (u′ = M⁻¹∇ℓq′) instead of the reference's two (hamiltonian.jl:278, :103).
    Same map, different rounding: identical trees, energies within 1e-11, positions within 1e-12 over 40 transitions of
    ≈ 37 leapfrogs at a fixed step size.  (With dual averaging ON the two runs drift apart much faster — 1e-6 after 40
    transitions here: the adaptation feeds last-place differences of the acceptance rate back into ϵ with a gain above one.
    That sensitivity belongs to the algorithm, whatever the arithmetic; it is why the tolerance tests fix ϵ.)"""
    D, C = 60, 4
    rng = np.random.default_rng(2)
    A = rng.normal(size=(D, D)); Minv = np.linalg.inv(A.T @ A / D + 0.2 * np.eye(D))
    prec = 1 / (rng.normal(size=D) ** 2 + 0.3)
    params = ol.target_params_blob(ol.TARGET_DIAG_NORMAL, D, mu=rng.normal(size=D), prec=prec)
    runs = {}
    for products in (1, 2):
        o = ol.Oracle(D, C, target=ol.TARGET_DIAG_NORMAL, params=params, metric=ol.METRIC_DENSE, seed=17, threads=4)
        o.set_dense_products(products)
        o.set_metric_dense(Minv)
        o.init(); o.find_initial_stepsize()
        eps_found = o.stepsize()
        o.set_stepsize(0.2)
        runs[products] = (eps_found, o.run(40))
    assert np.array_equal(runs[1][0], runs[2][0])            # the search is the reference's single leapfrogs in both
    a, b = runs[1][1], runs[2][1]
    assert a["steps"].mean() > 20
    for k in ("depth", "steps", "term_left", "term_right", "directions"):
        assert np.array_equal(a[k], b[k]), k
    assert not np.array_equal(a["draws"], b["draws"])        # it IS a different rounding
    assert np.abs(a["pi"] - b["pi"]).max() < 1e-11
    assert np.allclose(a["draws"], b["draws"], rtol=1e-12, atol=1e-12)


def test_sequential_sums_flavour_stays_next_to_the_abi_order():
    """oracle/mathops.hpp sequential_sums: every dot and matrix-vector product as a plain left-to-right loop with
    separately rounded products (the flavour tests/test_gpu_tolerance.py holds the device against)."""
    D, C = 300, 3
    runs = {}
    try:
        for seq in (False, True):
            ol.set_sequential_sums(seq)
            o = ol.Oracle(D, C, seed=23, threads=3)
            o.init(); o.set_stepsize(0.3)
            runs[seq] = o.run(12)
    finally:
        ol.set_sequential_sums(False)
    a, b = runs[False], runs[True]
    for k in ("depth", "steps", "term_left", "term_right", "directions"):
        assert np.array_equal(a[k], b[k]), k
    assert not np.array_equal(a["pi"], b["pi"])
    assert np.abs(a["pi"] - b["pi"]).max() < 1e-10
    assert np.allclose(a["draws"], b["draws"], rtol=1e-11, atol=1e-12)
