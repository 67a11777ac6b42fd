"""Randomised parity sweep: HIP path (through the C ABI) against the CPU oracle, bit for bit, over random dimensions,
targets, metrics, depths, step sizes and stage schedules.  `python tools/fuzz_parity.py [seconds] [seed]` on a GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from __graft_entry__ import load_package
pkg = load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); ncase = 0; nfail = 0; nrun = 0; ntrans = 0; nleap = 0
while time.time() - t0 < budget:
    kind = rng.choice(["std", "diag", "tridiag", "funnel", "logistic", "densenormal", "std_dense", "tridiag_dense", "divergent"])
    dense = kind.endswith("_dense")
    D = int(rng.choice([1, 2, 3, 5, 17, 30, 63, 64, 65, 100, 128, 129, 200, 256, 257, 400, 512, 700, 1000, 1024])) if not dense else int(rng.choice([2, 5, 17, 40, 64, 90, 90, 90, 520, 1000]))
    if kind in ("logistic",): D = int(rng.choice([2, 6, 20, 70]))
    if kind in ("densenormal",): D = int(rng.choice([2, 5, 12, 40]))
    if kind == "funnel": D = max(D, 2)
    C = int(rng.integers(1, 7))
    md = int(rng.choice([1, 2, 3, 5, 8, 10]))
    seed = int(rng.integers(0, 2**31)); off = int(rng.integers(0, 1000))
    params = None; target = ol.TARGET_STD_NORMAL
    if kind == "diag":
        target = ol.TARGET_DIAG_NORMAL; params = ol.target_params_blob(target, D, mu=rng.normal(size=D), prec=np.exp(rng.normal(size=D)))
    elif kind.startswith("tridiag"):
        target = ol.TARGET_TRIDIAG_NORMAL; params = ol.target_params_blob(target, D, diag=np.full(D, 2.0 + rng.random()), off=np.full(max(D - 1, 0), -0.9 * rng.random()))
    elif kind == "funnel":
        target = ol.TARGET_FUNNEL
    elif kind == "divergent":
        target = ol.TARGET_ALWAYS_DIVERGENT
    elif kind == "logistic":
        target = ol.TARGET_LOGISTIC; N = int(rng.choice([10, 64, 150, 333, 2048, 2049, 4500]))   # 2049+: more than one block of observations
        X = rng.normal(size=(N, D)) / 2; y = (rng.random(N) < 0.5).astype(float)
        params = ol.target_params_blob(target, D, X=X, y=y)
    elif kind == "densenormal":
        target = ol.TARGET_DENSE_NORMAL; A = rng.normal(size=(D, D)); P = A @ A.T / D + np.eye(D)
        params = ol.target_params_blob(target, D, mu=rng.normal(size=D), P=P)
    kw = dict(target=target, seed=seed, max_depth=md, chain_offset=off, metric=ol.METRIC_DENSE if dense else ol.METRIC_DIAG)
    # engine choice (read at context creation): the round engines, normally chosen from 128 chains up, and their K3 variants
    env = {}
    if dense and rng.random() < 0.4: env = {"DHMC_DENSE": "rounds=1,k3_block=%d" % int(rng.random() < 0.5)}
    if kind == "logistic" and rng.random() < 0.5: env = {"DHMC_LOGISTIC_ROUNDS": "1"}
    os.environ.update(env)
    try:
        dev = pkg.DeviceContext(D, C, target_params=params, **kw)
    finally:
        for k_ in env: os.environ.pop(k_, None)
    ora = ol.Oracle(D, C, params=params, threads=4, **kw)
    desc = f"{kind} D={D} C={C} max_depth={md} seed={seed} offset={off} env={env}"
    if os.environ.get("FUZZ_VERBOSE") == "1": print("case", desc, file=sys.stderr, flush=True)
    try:
        q0 = None if rng.random() < 0.5 or kind == "divergent" else rng.normal(size=(C, D))
        if kind == "divergent": q0 = np.zeros((C, D))
        a = dev.init(q0, allow_failure=True); b = ora.init(q0, allow_failure=True)
        assert a == b, "init rc"
        if dense:
            A = rng.normal(size=(D, D)); S = A @ A.T / D + np.eye(D)
            dev.set_metric_dense(S); ora.set_metric_dense(S)
        elif rng.random() < 0.5:
            m = np.exp(rng.normal(size=(C, D)) * 0.5); dev.set_metric_diag(m); ora.set_metric_diag(m)
        if rng.random() < 0.6:
            a = dev.find_initial_stepsize(allow_failure=True); b = ora.find_initial_stepsize(allow_failure=True)
            assert a == b, "search rc"
        else:
            e = float(np.exp(rng.normal()) * 0.3 / max(D, 1) ** 0.25); dev.set_stepsize(e); ora.set_stepsize(e)
        assert np.array_equal(dev.stepsize(), ora.stepsize(), equal_nan=True), "eps after search"
        if np.isnan(dev.stepsize()).any() or dev.status().any():
            ncase += 1; continue
        for _ in range(int(rng.integers(1, 4))):
            n = int(rng.integers(1, 25 if D < 500 or not dense else 4)); adapt = rng.random() < 0.6
            x = dev.run(n, da={} if adapt else None, allow_failure=True); y_ = ora.run(n, da={} if adapt else None, allow_failure=True)
            for k in x:
                if not np.array_equal(x[k], y_[k], equal_nan=True) and os.environ.get("FUZZ_VERBOSE"):   # where and by how much
                    bad = np.argwhere(~((x[k] == y_[k]) | (np.isnan(x[k].astype(float)) & np.isnan(y_[k].astype(float)))))
                    i = tuple(bad[0]); c_, t_ = int(i[0]), int(i[1])
                    print(f"  first difference in {k} at {i} of {x[k].shape} ({len(bad)} entries differ): device {x[k][i]!r} oracle {y_[k][i]!r}", file=sys.stderr)
                    for f_ in ("depth", "steps", "term_left", "term_right", "acceptance_rate", "pi", "logdensities", "eps"):
                        if f_ in x: print(f"    chain {c_}: {f_} device {x[f_][c_][max(t_ - 1, 0):t_ + 2]!r} oracle {y_[f_][c_][max(t_ - 1, 0):t_ + 2]!r}", file=sys.stderr)
                assert np.array_equal(x[k], y_[k], equal_nan=True), f"field {k}"
            nrun += 1; ntrans += n * C; nleap += int(x["steps"].sum())
            assert np.array_equal(dev.stepsize(), ora.stepsize(), equal_nan=True), "eps"
            assert np.array_equal(dev.status(), ora.status()), "status"
            if dev.status().any(): break
            if not dense and adapt and n >= 3 and rng.random() < 0.5:
                dev.update_metric_diag(x["draws"]); ora.update_metric_diag(y_["draws"])
                assert np.array_equal(dev.metric_diag(), ora.metric_diag()), "metric"
    except AssertionError as ex:
        nfail += 1; print("MISMATCH", desc, "->", ex, flush=True)
    except Exception as ex:
        nfail += 1; print("ERROR", desc, "->", repr(ex), flush=True)
    ncase += 1
print(f"{ncase} random cases in {time.time() - t0:.0f} s: {nrun} compared runs, {ntrans} transitions, {nleap} leapfrog steps, {nfail} failures")
