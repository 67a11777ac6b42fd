"""Replay of the reference's deterministic tree tests (test/test_trees.jl) against the oracle.

These are the known-answer tests that pin the tree-doubling / multinomial logic (SURVEY.md
§8c items 1-3): direction bit order, the five DummyTrajectory shape tests, and the exhaustive
detailed-balance enumeration.
"""
import math

import numpy as np
import oracle_lib as ol


def _testl(z):  # test_trees.jl:106
    return -abs(z - 3) ** 2 * 0.1


def _testA(zs):  # test_trees.jl:109
    return sum(min(math.exp(_testl(z)), 1) for z in zs)


def test_directions():  # test_trees.jl:8-15
    assert ol.directions(0b110101, 6) == [True, False, True, False, True, True]


def test_dummy_adjacent_tree_full():  # test_trees.jl:114-124
    o, logp, vis = ol.dummy_adjacent_tree(0, 0, 2, True)
    assert o.valid and (o.zeta_first, o.zeta_last) == (1, 4)
    assert np.isclose(np.exp(logp).sum(), 1)
    assert list(vis) == [1, 2, 3, 4]
    assert not o.tau_flag
    assert np.isclose(o.v_a, _testA(vis)) and o.v_s == 4
    assert o.zlast == o.ilast == 4
    assert o.assertion_failures == 0


def test_dummy_adjacent_tree_turning():  # test_trees.jl:126-133
    o, _, vis = ol.dummy_adjacent_tree(0, 0, 3, True, turning=range(5, 8))
    assert list(vis) == [1, 2, 3, 4, 5, 6]
    assert not o.valid and (o.inv_left, o.inv_right) == (5, 6)
    assert np.isclose(o.v_a, _testA(vis)) and o.v_s == 6


def test_dummy_adjacent_tree_divergent():  # test_trees.jl:135-142
    o, _, vis = ol.dummy_adjacent_tree(0, 0, 3, True, divergent=range(5, 8))
    assert list(vis) == [1, 2, 3, 4, 5]
    assert not o.valid and (o.inv_left, o.inv_right) == (5, 5)
    assert np.isclose(o.v_a, _testA(range(1, 6))) and o.v_s == 5


def test_dummy_adjacent_tree_full_backward():  # test_trees.jl:144-154
    o, logp, vis = ol.dummy_adjacent_tree(0, 0, 3, False)
    assert o.valid and (o.zeta_first, o.zeta_last) == (-8, -1)
    assert np.isclose(np.exp(logp).sum(), 1)
    assert list(vis) == [-i for i in range(1, 9)]
    assert not o.tau_flag
    assert np.isclose(o.v_a, _testA(vis)) and o.v_s == 8
    assert o.zlast == o.ilast == -8
    assert o.assertion_failures == 0


def test_dummy_sampled_tree():  # test_trees.jl:156-165
    o, logp, vis = ol.dummy_sample_trajectory(0, 3, 0b101)
    assert list(vis) == [1, -1, -2, 2, 3, 4, 5]
    assert (o.zeta_first, o.zeta_last) == (-2, 5)
    assert np.isclose(np.exp(logp).sum(), 1)
    assert (o.inv_left, o.inv_right) == (1, 0)  # REACHED_MAX_DEPTH
    assert np.isclose(o.v_a, _testA(vis)) and o.v_s == 7
    assert o.assertion_failures == 0


# ---- detailed balance (test_trees.jl:171-262) -------------------------------------------

def _logaddexp(a, b):
    return float(np.logaddexp(a, b))


def visited_log_probabilities(z, depth, **kw):  # test_trees.jl:192-199
    acc = {}
    for flags in range(2 ** depth):
        o, logp, _ = ol.dummy_sample_trajectory(z, depth, flags, **kw)
        assert o.assertion_failures == 0
        for zz, lp in zip(range(o.zeta_first, o.zeta_last + 1), logp):
            acc[zz] = _logaddexp(acc[zz], lp) if zz in acc else lp
    D = math.log(0.5) * depth
    return {k: v + D for k, v in acc.items()}


def transition_log_probability(z, z1, depth, **kw):  # test_trees.jl:205-216
    p = -math.inf
    for flags in range(2 ** depth):
        o, logp, _ = ol.dummy_sample_trajectory(z, depth, flags, **kw)
        if o.zeta_first <= z1 <= o.zeta_last:
            p = _logaddexp(p, logp[z1 - o.zeta_first])
    return p + depth * math.log(0.5)


def test_transition_calculations_consistency():  # test_trees.jl:218-225
    for z1, pi in visited_log_probabilities(9, 5).items():
        assert np.isclose(pi, transition_log_probability(9, z1, 5))


def check_detailed_balance(z, depth, **kw):  # test_trees.jl:239-246
    atol = math.sqrt(np.finfo(float).eps)
    lz = _testl(z)
    for z1, pi in visited_log_probabilities(z, depth, **kw).items():
        pi1 = transition_log_probability(z1, z, depth, **kw)
        assert abs((pi + lz) - (pi1 + _testl(z1))) <= atol, (z, z1, depth, kw)


def test_detailed_balance():  # test_trees.jl:248-262, all 22 (trajectory, depth) cases
    for d in range(1, 6):
        check_detailed_balance(0, d)
    for d in range(1, 6):
        check_detailed_balance(3, d, turning=range(1, 3))
    for d in range(1, 7):
        check_detailed_balance(3, d, divergent=range(10, 12))
    for d in range(1, 7):
        check_detailed_balance(3, d, divergent=range(10, 13), turning=range(-3, -1))
