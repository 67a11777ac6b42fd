"""Generates tests/golden/*.npz from the CPU oracle (deterministic-math flavour).

The reference itself cannot run here (Julia is not installed), so these are outputs of the
oracle — the C++ restatement of the reference that the reference's own known-answer tests pin
(tests/test_oracle_*.py).  Each fixture holds the inputs (configuration, seeds) and every
output of a short warmup + inference run; `tests/test_golden.py` replays them against the
oracle (regression pin) and, on the GPU box, against the HIP path (bit for bit).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

CASES = {
    # name: (D, chains, target, target kwargs, seed, max_depth, [(n, adapt, metric_update)...])
    "stdnormal_d100": (100, 3, ol.TARGET_STD_NORMAL, {}, 0x23EF614D, 10, [(20, True, False), (25, True, True), (10, False, False)]),
    "stdnormal_d1000": (1000, 2, ol.TARGET_STD_NORMAL, {}, 7, 10, [(10, True, False), (6, False, False)]),
    "diagnormal_d5": (5, 4, ol.TARGET_DIAG_NORMAL, dict(mu=np.ones(5), prec=np.array([1.0, 4.0, 0.25, 1.0, 9.0])), 11, 10,
                      [(30, True, False), (25, True, True), (20, False, False)]),
    "tridiag_d40": (40, 3, ol.TARGET_TRIDIAG_NORMAL, dict(diag=np.full(40, 5 / 3.0), off=np.full(40, -2 / 3.0)), 5, 10,
                    [(25, True, False), (10, False, False)]),
    "funnel_d30": (30, 6, ol.TARGET_FUNNEL, {}, 4, 10, [(40, True, False), (25, True, True), (25, False, False)]),
    "shallow_d20_maxdepth3": (20, 3, ol.TARGET_STD_NORMAL, {}, 9, 3, [(20, True, False), (10, False, False)]),
    "logistic_n150_d6": (6, 3, ol.TARGET_LOGISTIC, "logistic", 21, 10, [(25, True, True), (15, False, False)]),
    # dense metric: correlated normal (tridiagonal precision, rho = 0.5) with the perfect metric M⁻¹ = Σ — the reference's
    # recurrence (two M⁻¹ products per leapfrog; the fixture of rounds 1-2) and the one-product recurrence (round 3)
    "dense_tridiag_d12": (12, 3, ol.TARGET_TRIDIAG_NORMAL, dict(diag=np.r_[4 / 3.0, np.full(10, 5 / 3.0), 4 / 3.0], off=np.full(12, -2 / 3.0)),
                          13, 10, [(25, True, False), (15, False, False)], "dense2"),
    "dense_tridiag_d12_one_product": (12, 3, ol.TARGET_TRIDIAG_NORMAL, dict(diag=np.r_[4 / 3.0, np.full(10, 5 / 3.0), 4 / 3.0], off=np.full(12, -2 / 3.0)),
                                      13, 10, [(25, True, False), (15, False, False)], "dense1"),
}


def _logistic_data():
    rng = np.random.default_rng(2024)
    X = rng.normal(size=(150, 6)) / 2
    y = (rng.random(150) < 1 / (1 + np.exp(-X @ rng.normal(size=6)))).astype(float)
    return dict(X=X, y=y)


def run_case(engine_factory, spec):
    D, C, target, tkw, seed, max_depth, stages = spec[:7]
    dense = len(spec) > 7
    if tkw == "logistic":
        tkw = _logistic_data()
    params = ol.target_params_blob(target, D, **tkw)
    eng = engine_factory(D, C, target, params, seed, max_depth, ol.METRIC_DENSE if dense else ol.METRIC_DIAG)
    out = {}
    if dense:
        P = np.diag(tkw["diag"]) + np.diag(tkw["off"][:D - 1], 1) + np.diag(tkw["off"][:D - 1], -1)
        eng.set_dense_products(int(spec[7][-1]))
        eng.set_metric_dense(np.linalg.inv(P))
    eng.init()
    q, lq, g = eng.position()
    out["init_q"], out["init_lq"], out["init_grad"] = q, lq, g
    eng.find_initial_stepsize()
    out["search_eps"] = eng.stepsize()
    for si, (n, adapt, metric) in enumerate(stages):
        r = eng.run(n, da={} if adapt else None)
        for k, v in r.items():
            out[f"s{si}_{k}"] = v
        out[f"s{si}_eps_after"] = eng.stepsize()
        if metric and not dense:
            eng.update_metric_diag(r["draws"])
            out[f"s{si}_metric_after"] = eng.metric_diag()
    return out


def oracle_factory(D, C, target, params, seed, max_depth, metric):
    return ol.Oracle(D, C, target=target, params=params, seed=seed, max_depth=max_depth, metric=metric)


if __name__ == "__main__":
    for name, spec in CASES.items():
        out = run_case(oracle_factory, spec)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, f"{os.path.getsize(path) / 1024:.0f} KiB", {k: v.shape for k, v in list(out.items())[:3]})
