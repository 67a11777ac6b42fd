"""ctypes harness over oracle/liboracle.so (the CPU restatement of the reference hot path).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Builds the library with `make -C oracle` when it is missing or stale.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

# constants of include/dhmc.h
OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_UNSUPPORTED, ERR_CHAIN_FAILURE, ERR_NO_DEVICE = range(6)
ST_NONFINITE_POSITION, ST_INVALID_INITIAL, ST_STEPSIZE_SEARCH_FAILED, ST_NONFINITE_START_DENSITY = 1, 2, 4, 8
TARGET_STD_NORMAL, TARGET_DIAG_NORMAL, TARGET_TRIDIAG_NORMAL, TARGET_FUNNEL, TARGET_LOGISTIC, TARGET_ALWAYS_DIVERGENT, TARGET_DENSE_NORMAL = range(7)
METRIC_DIAG, METRIC_DENSE = 0, 1


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("dim", C.c_int32), ("chains", C.c_int32),
                ("chain_offset", C.c_int32), ("metric", C.c_int32), ("target", C.c_int32),
                ("target_params", C.c_void_p), ("target_params_bytes", C.c_uint64),
                ("max_depth", C.c_int32), ("dense_per_chain", C.c_int32), ("min_delta", C.c_double),
                ("seed", C.c_uint64)]


class StepsizeSearch(C.Structure):
    _fields_ = [("initial_eps", C.c_double), ("log_threshold", C.c_double),
                ("maxiter_crossing", C.c_int32), ("reserved", C.c_int32)]


class DualAveraging(C.Structure):
    _fields_ = [("delta", C.c_double), ("gamma", C.c_double), ("kappa", C.c_double),
                ("t0", C.c_int32), ("init", C.c_int32), ("finalize", C.c_int32),
                ("reserved", C.c_int32)]


class Outputs(C.Structure):
    _fields_ = [("on_device", C.c_int32), ("reserved", C.c_int32), ("draws", C.c_void_p),
                ("logdensities", C.c_void_p), ("eps", C.c_void_p), ("pi", C.c_void_p),
                ("acceptance_rate", C.c_void_p), ("steps", C.c_void_p), ("term_left", C.c_void_p),
                ("term_right", C.c_void_p), ("depth", C.c_void_p), ("directions", C.c_void_p)]


class DummyOut(C.Structure):
    _fields_ = [("valid", C.c_int32), ("depth", C.c_int32), ("tau_flag", C.c_int32),
                ("assertion_failures", C.c_int32), ("inv_left", C.c_int64), ("inv_right", C.c_int64),
                ("zeta_first", C.c_int64), ("zeta_last", C.c_int64), ("tau_first", C.c_int64),
                ("tau_last", C.c_int64), ("zlast", C.c_int64), ("ilast", C.c_int64), ("v_s", C.c_int64),
                ("omega", C.c_double), ("v_a", C.c_double), ("n_logp", C.c_int64), ("n_visited", C.c_int64)]


OUTPUT_FIELDS = [("draws", np.float64), ("logdensities", np.float64), ("eps", np.float64),
                 ("pi", np.float64), ("acceptance_rate", np.float64), ("steps", np.int64),
                 ("term_left", np.int64), ("term_right", np.int64), ("depth", np.int32),
                 ("directions", np.uint32)]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    srcs += [os.path.join(ROOT, "include", f) for f in ("dhmc.h", "dhmc_detmath.h", "dhmc_detmath_tables.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if _stale():
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_unit_wave_dot.restype = C.c_double
        _lib.oracle_unit_acceptance.restype = C.c_double
        _lib.oracle_unit_logdensity.restype = C.c_double
        _lib.oracle_unit_rand_bool_logprob.restype = C.c_int64
        _lib.oracle_unit_leapfrog.restype = C.c_uint32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def target_params_blob(target, D, **kw):
    """Host blob for dhmc_config.target_params (layout in include/dhmc.h)."""
    if target == TARGET_DIAG_NORMAL:
        return np.concatenate([np.asarray(kw["mu"], np.float64), np.asarray(kw["prec"], np.float64)])
    if target == TARGET_TRIDIAG_NORMAL:
        off = np.zeros(D)
        o = np.asarray(kw["off"], np.float64)
        off[:len(o)] = o
        return np.concatenate([np.asarray(kw["diag"], np.float64), off])
    if target == TARGET_DENSE_NORMAL:
        return np.concatenate([np.asarray(kw["mu"], np.float64), np.ascontiguousarray(kw["P"], np.float64).ravel()])
    if target == TARGET_LOGISTIC:
        X = np.ascontiguousarray(kw["X"], np.float64); y = np.ascontiguousarray(kw["y"], np.float64)
        return np.concatenate([np.array([X.shape[0]], np.int64).view(np.float64), X.ravel(), y])
    return None


def make_config(D, chains, target=TARGET_STD_NORMAL, seed=0x23EF614D, max_depth=10,
                min_delta=-1000.0, chain_offset=0, metric=METRIC_DIAG, device=0, params=None, dense_per_chain=False):
    cfg = Config()
    cfg.device, cfg.dim, cfg.chains, cfg.chain_offset = device, D, chains, chain_offset
    cfg.metric, cfg.target, cfg.max_depth, cfg.min_delta, cfg.seed = metric, target, max_depth, min_delta, seed
    cfg.dense_per_chain = int(bool(dense_per_chain))
    if params is not None:
        params = np.ascontiguousarray(params)
        cfg.target_params = params.ctypes.data
        cfg.target_params_bytes = params.nbytes
    cfg._keep = params
    return cfg


def alloc_outputs(C_, N, D, fields=None):
    arrs = {}
    for name, dt in OUTPUT_FIELDS:
        if fields is not None and name not in fields:
            continue
        shape = (C_, N, D) if name == "draws" else (C_, N)
        arrs[name] = np.zeros(shape, dt)
    return arrs


def outputs_struct(arrs, on_device=0):
    o = Outputs()
    o.on_device = on_device
    for name, _ in OUTPUT_FIELDS:
        a = arrs.get(name)
        setattr(o, name, a.ctypes.data if a is not None else None)
    return o


def set_sequential_sums(on):
    """Process-wide: every sum of the oracle as a plain left-to-right loop instead of the ABI's order (oracle/mathops.hpp
    sequential_sums) — the second flavour of the tolerance tests.  Reset it to False when done."""
    lib().oracle_set_sequential_sums(int(bool(on)))


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error code {code}")
        self.code = code


class Oracle:
    """Multi-chain oracle context with the same call surface as the dhmc C ABI."""

    def __init__(self, D, chains, target=TARGET_STD_NORMAL, det=True, threads=1, **kw):
        params = kw.pop("params", None)
        self.cfg = make_config(D, chains, target=target, params=params, **kw)
        self.D, self.C = D, chains
        self.h = C.c_void_p()
        rc = lib().oracle_create(C.byref(self.cfg), int(det), C.byref(self.h))
        if rc != OK:
            raise OracleError(rc)
        lib().oracle_set_threads(self.h, threads)
        # a shared dense metric runs the one-product recurrence by default, as the device library does (include/dhmc.h
        # dhmc_set_dense_products); DHMC_DENSE="products=2" flips both defaults
        if self.cfg.metric == METRIC_DENSE and not self.cfg.dense_per_chain and "products=2" not in os.environ.get("DHMC_DENSE", "").split(","):
            self.set_dense_products(1)

    def close(self):
        if getattr(self, "h", None):
            lib().oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc, allow=(OK,)):
        if rc not in allow:
            raise OracleError(rc)
        return rc

    def init(self, q0=None, allow_failure=False):
        q0 = None if q0 is None else np.ascontiguousarray(q0, np.float64)
        return self._chk(lib().oracle_init(self.h, _p(q0)), (OK, ERR_CHAIN_FAILURE) if allow_failure else (OK,))

    def position(self):
        q = np.zeros((self.C, self.D)); lq = np.zeros(self.C); g = np.zeros((self.C, self.D))
        self._chk(lib().oracle_get_position(self.h, _p(q), _p(lq), _p(g)))
        return q, lq, g

    def set_metric_diag(self, minv):
        minv = np.ascontiguousarray(minv, np.float64)
        self._chk(lib().oracle_set_metric_diag(self.h, _p(minv), int(minv.ndim == 2)))

    def set_dense_products(self, products):
        self._chk(lib().oracle_set_dense_products(self.h, int(products)))

    def set_metric_dense(self, minv):
        minv = np.ascontiguousarray(minv, np.float64)
        self._chk(lib().oracle_set_metric_dense(self.h, _p(minv)))

    def update_metric_dense(self, draws, lam):
        draws = np.ascontiguousarray(draws, np.float64)
        self._chk(lib().oracle_update_metric_dense(self.h, _p(draws), C.c_int64(draws.shape[1]), C.c_double(lam)))

    def metric_dense(self, chain=0):
        M = np.zeros((self.D, self.D)); W = np.zeros((self.D, self.D))
        self._chk(lib().oracle_get_metric_dense_chain(self.h, C.c_int32(chain), _p(M), _p(W)))
        return M, W

    def metric_dense_W(self):
        return self.metric_dense()[1]

    def metric_diag(self):
        m = np.zeros((self.C, self.D))
        self._chk(lib().oracle_get_metric_diag(self.h, _p(m)))
        return m

    def set_stepsize(self, eps):
        eps = np.ascontiguousarray(np.atleast_1d(eps), np.float64)
        self._chk(lib().oracle_set_stepsize(self.h, _p(eps), int(eps.size == self.C)))

    def stepsize(self):
        e = np.zeros(self.C)
        self._chk(lib().oracle_get_stepsize(self.h, _p(e)))
        return e

    def status(self):
        s = np.zeros(self.C, np.uint32)
        self._chk(lib().oracle_get_status(self.h, _p(s)))
        return s

    def find_initial_stepsize(self, initial_eps=0.1, log_threshold=float(np.log(0.8)), maxiter_crossing=400,
                              allow_failure=False):
        p = StepsizeSearch(initial_eps, log_threshold, maxiter_crossing, 0)
        return self._chk(lib().oracle_find_initial_stepsize(self.h, C.byref(p)),
                         (OK, ERR_CHAIN_FAILURE) if allow_failure else (OK,))

    def run(self, N, da=None, fields=None, allow_failure=False):
        arrs = alloc_outputs(self.C, N, self.D, fields)
        o = outputs_struct(arrs)
        dap = None
        if da is not None:
            d = dict(delta=0.8, gamma=0.05, kappa=0.75, t0=10, init=1, finalize=1)
            d.update(da)
            dap = DualAveraging(d["delta"], d["gamma"], d["kappa"], d["t0"], d["init"], d["finalize"], 0)
        rc = lib().oracle_run(self.h, C.c_int64(N), C.byref(dap) if dap is not None else None, C.byref(o))
        self._chk(rc, (OK, ERR_CHAIN_FAILURE) if allow_failure else (OK,))
        return arrs

    def update_metric_diag(self, draws, lam=0.0):
        draws = np.ascontiguousarray(draws, np.float64)
        self._chk(lib().oracle_update_metric_diag(self.h, _p(draws), C.c_int64(draws.shape[1]), C.c_double(lam)))

    def metric_window_begin(self):
        self._chk(lib().oracle_metric_window_begin(self.h))

    def update_metric_diag_window(self, lam=0.0):
        self._chk(lib().oracle_update_metric_diag_window(self.h, C.c_double(lam)))

    def leapfrog_trajectory(self, eps, first, last, p=None, momentum_index=0, allow_failure=False):
        npos = last - first + 1
        out = dict(delta=np.zeros((self.C, npos)), logdensity=np.zeros((self.C, npos)),
                   range=np.zeros((self.C, 2), np.int32), status=np.zeros(self.C, np.uint32),
                   q=np.zeros((self.C, npos, self.D)), p=np.zeros((self.C, npos, self.D)))
        if p is not None:
            p = np.ascontiguousarray(np.broadcast_to(p, (self.C, self.D)), np.float64)
        rc = lib().oracle_leapfrog_trajectory(self.h, C.c_double(eps), C.c_int32(first), C.c_int32(last),
                                              C.c_uint32(momentum_index), _p(p), _p(out["delta"]), _p(out["logdensity"]),
                                              _p(out["q"]), _p(out["p"]), _p(out["range"]), _p(out["status"]))
        self._chk(rc, (OK, ERR_CHAIN_FAILURE) if allow_failure else (OK,))
        return out

    def explore_log_acceptance_ratios(self, eps, n_momenta=20, ps=None, momentum_index=0, allow_failure=False):
        eps = np.ascontiguousarray(np.atleast_1d(eps), np.float64)
        if ps is not None:
            ps = np.ascontiguousarray(ps, np.float64)
            if ps.ndim == 2:
                ps = np.ascontiguousarray(np.broadcast_to(ps, (self.C,) + ps.shape))
            n_momenta = ps.shape[1]
        out = np.zeros((self.C, n_momenta, eps.size))
        status = np.zeros(self.C, np.uint32)
        rc = lib().oracle_explore_log_acceptance_ratios(self.h, _p(eps), C.c_int32(eps.size), C.c_int32(n_momenta),
                                                        C.c_uint32(momentum_index), _p(ps), _p(out), _p(status))
        self._chk(rc, (OK, ERR_CHAIN_FAILURE) if allow_failure else (OK,))
        return out

    def da_state(self):
        mu = np.zeros(self.C); m = np.zeros(self.C, np.int64); hb = np.zeros(self.C)
        le = np.zeros(self.C); leb = np.zeros(self.C)
        lib().oracle_get_da_state(self.h, _p(mu), _p(m), _p(hb), _p(le), _p(leb))
        return dict(mu=mu, m=m, Hbar=hb, logeps=le, logeps_bar=leb)


# ---- unit hooks --------------------------------------------------------------------------

def detmath(kind, x, y=None):
    x = np.ascontiguousarray(x, np.float64)
    y = None if y is None else np.ascontiguousarray(y, np.float64)
    out = np.empty_like(x)
    lib().oracle_unit_detmath(kind, C.c_int64(x.size), _p(x), _p(y), _p(out))
    return out


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
    lib().oracle_unit_philox(_p(c), _p(k), _p(o))
    return o


def wave_dot(a, b):
    a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
    return lib().oracle_unit_wave_dot(_p(a), _p(b), C.c_int(a.size))


def directions(flags, n):
    o = (C.c_int * n)()
    lib().oracle_unit_directions(C.c_uint32(flags), n, o)
    return [bool(v) for v in o]


def _dummy_call(fn, turning, divergent, *args, ell_c=3.0, ell_a=0.1):
    t = np.asarray(sorted(turning), np.int64); d = np.asarray(sorted(divergent), np.int64)
    o = DummyOut(); cap = 1 << 12
    logp = np.zeros(cap); vis = np.zeros(cap, np.int64)
    fn(C.c_double(ell_c), C.c_double(ell_a), _p(t), len(t), _p(d), len(d), *args, C.byref(o),
       _p(logp), C.c_int64(cap), _p(vis), C.c_int64(cap))
    return o, logp[:o.n_logp].copy(), vis[:o.n_visited].copy()


def dummy_adjacent_tree(z, i, depth, fwd, turning=(), divergent=()):
    return _dummy_call(lib().oracle_unit_dummy_adjacent_tree, turning, divergent,
                       C.c_int64(z), C.c_int64(i), C.c_int(depth), C.c_int(int(fwd)))


def dummy_sample_trajectory(z, max_depth, flags, turning=(), divergent=()):
    return _dummy_call(lib().oracle_unit_dummy_sample_trajectory, turning, divergent,
                       C.c_int64(z), C.c_int(max_depth), C.c_uint32(flags))


def combine_turn(x, y):
    """x, y: arrays [5][D] = (p₋, p♯₋, p₊, p♯₊, ρ).  Returns (turning, rho)."""
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    D = x.shape[1]; rho = np.zeros(D)
    t = lib().oracle_unit_combine_turn(D, _p(x), _p(y), _p(rho))
    return bool(t), rho


def acceptance(deltas, is_initial, det=True):
    d = np.ascontiguousarray(deltas, np.float64); ii = np.ascontiguousarray(is_initial, np.int32)
    return lib().oracle_unit_acceptance(int(det), len(d), _p(d), _p(ii))


def rand_bool_logprob(logprob, n, seed=1, det=True):
    used = C.c_int64()
    cnt = lib().oracle_unit_rand_bool_logprob(int(det), C.c_double(logprob), C.c_uint64(seed), C.c_int64(n), C.byref(used))
    return cnt, used.value


def logdensity(lq, p, minv):
    p = np.ascontiguousarray(p, np.float64); minv = np.ascontiguousarray(minv, np.float64)
    return lib().oracle_unit_logdensity(len(p), C.c_double(lq), _p(p), _p(minv))


def leapfrog(cfg, minv, q0, p0, eps, n, det=True):
    D = cfg.dim
    minv = np.ascontiguousarray(minv, np.float64); q0 = np.ascontiguousarray(q0, np.float64)
    p0 = np.ascontiguousarray(p0, np.float64)
    qs = np.zeros((n, D)); ps = np.zeros((n, D)); pis = np.zeros(n); lqs = np.zeros(n)
    st = lib().oracle_unit_leapfrog(C.byref(cfg), int(det), _p(minv), _p(q0), _p(p0), C.c_double(eps), n,
                                    _p(qs), _p(ps), _p(pis), _p(lqs))
    return qs, ps, pis, lqs, st


def find_initial_stepsize_linear(slope, intercept=0.0, initial_eps=0.1, log_threshold=float(np.log(0.8)), maxiter=400):
    eps = C.c_double()
    rc = lib().oracle_unit_find_initial_stepsize_linear(C.c_double(slope), C.c_double(intercept), C.c_double(initial_eps),
                                                        C.c_double(log_threshold), maxiter, C.byref(eps))
    return rc, eps.value


def da_init(eps, det=True):
    st = np.zeros(5)
    lib().oracle_unit_da_init(int(det), C.c_double(eps), _p(st))
    return st


def da_adapt(st, a, delta=0.8, gamma=0.05, kappa=0.75, t0=10, det=True):
    st = np.array(st, np.float64)
    lib().oracle_unit_da_adapt(int(det), C.c_double(delta), C.c_double(gamma), C.c_double(kappa), t0, _p(st), C.c_double(a))
    return st


def rand_p_dense(minv, n, seed=1):
    minv = np.ascontiguousarray(minv, np.float64); D = minv.shape[0]
    out = np.zeros((n, D)); W = np.zeros((D, D))
    rc = lib().oracle_unit_rand_p_dense(D, _p(minv), C.c_uint64(seed), n, _p(out), _p(W))
    if rc:
        raise ValueError("not positive definite")
    return out, W


def summarize_tree_statistics(pi, acceptance_rate, term_left, term_right, depth):
    """Oracle flavour of dhmc_summarize_tree_statistics (oracle/diagnostics.hpp): returns (summary dict, ebfmi [C])."""
    class S(C.Structure):
        _fields_ = [("n", C.c_int64), ("a_mean", C.c_double), ("a_quantiles", C.c_double * 5), ("max_depth", C.c_int64),
                    ("divergence", C.c_int64), ("turning", C.c_int64), ("depth_counts", C.c_int64 * 33)]
    a = [np.ascontiguousarray(x, t) for x, t in zip((pi, acceptance_rate, term_left, term_right, depth),
                                                     (np.float64, np.float64, np.int64, np.int64, np.int32))]
    Cn, N = a[0].shape
    out = S(); eb = np.zeros(Cn)
    rc = lib().oracle_summarize_tree_statistics(*[C.c_void_p(x.ctypes.data) for x in a], C.c_int64(Cn), C.c_int64(N),
                                                C.byref(out), C.c_void_p(eb.ctypes.data))
    assert rc == 0
    dc = list(out.depth_counts)
    last = max((i for i, v in enumerate(dc) if v), default=-1)
    return dict(N=int(out.n), a_mean=float(out.a_mean), a_quantiles=list(out.a_quantiles),
                termination_counts=dict(max_depth=int(out.max_depth), divergence=int(out.divergence), turning=int(out.turning)),
                depth_counts=dc[:last + 1]), eb
