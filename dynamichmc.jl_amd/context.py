"""Low-level multi-chain context: a thin object over the C ABI (one per GPU).

Mirrors the reference's WarmupState(Q, κ, ϵ) (src/mcmc.jl:72-79) for C chains at once and the
calls that act on it.  Failures the reference raises as DynamicHMCError / ArgumentError are
re-raised here under the same names.
"""
import ctypes as C

import numpy as np

from . import _abi as abi


class DynamicHMCError(Exception):
    """Counterpart of DynamicHMC.DynamicHMCError (src/utilities.jl:17-27): `message` plus a
    dict of debug information (here: the failing chains and their status words)."""

    def __init__(self, message, **debug_information):
        super().__init__(message)
        self.message = message
        self.debug_information = debug_information

    def __str__(self):
        s = f"DynamicHMC error: {self.message}"
        for k, v in self.debug_information.items():
            s += f"\n  {k} = {v}"
        return s


STATUS_MESSAGES = [
    (abi.ST_NONFINITE_POSITION, "Position vector has non-finite elements."),               # hamiltonian.jl:203
    (abi.ST_INVALID_INITIAL, "Invalid log posterior or non-finite gradient at the initial position."),  # :212-216
    (abi.ST_STEPSIZE_SEARCH_FAILED, "Initial stepsize search reached maximum number of iterations without crossing."),  # stepsize.jl:58
    (abi.ST_NONFINITE_START_DENSITY, "Starting point has non-finite density."),           # stepsize.jl:78
    (0x40000000, "Internal error: a bounded wait inside a kernel ran out (DHMC_ST_KERNEL_PROTOCOL, include/dhmc.h); not the model's fault."),
]


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a.data_ptr()))  # torch tensor


def _is_device(a):
    return not isinstance(a, np.ndarray) and hasattr(a, "data_ptr") and a.is_cuda


class _PinnedBlock:
    """A page-locked allocation of the library (dhmc_host_alloc), returned to a small pool when the last numpy view of it
    goes away."""
    _pool = {}          # nbytes -> [address, ...]
    _POOL_BYTES = 8 << 30
    _pooled = 0

    def __init__(self, nbytes):
        self.nbytes = nbytes
        free = _PinnedBlock._pool.get(nbytes)
        if free:
            self.addr = free.pop()
            _PinnedBlock._pooled -= nbytes
        else:
            p = C.c_void_p()
            rc = abi.lib().dhmc_host_alloc(C.byref(p), C.c_uint64(nbytes))
            if rc != abi.OK or not p.value:
                raise MemoryError(f"dhmc_host_alloc({nbytes}) failed")
            self.addr = p.value

    def __del__(self):
        try:
            if _PinnedBlock._pooled + self.nbytes <= _PinnedBlock._POOL_BYTES:
                _PinnedBlock._pool.setdefault(self.nbytes, []).append(self.addr)
                _PinnedBlock._pooled += self.nbytes
            else:
                abi.lib().dhmc_host_free(C.c_void_p(self.addr))
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array (uninitialised) in page-locked host memory — what dhmc_run's host outputs want (include/dhmc.h)."""
    dtype = np.dtype(dtype)
    if any(int(d) <= 0 for d in shape):
        return np.empty(shape, dtype)            # empty, or numpy's own ValueError for a negative dimension
    n = int(np.prod(shape)) * dtype.itemsize
    blk = _PinnedBlock(n)
    buf = (C.c_char * n).from_address(blk.addr)
    buf._dhmc_block = blk                      # keeps the block alive as long as any view of the buffer is
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class DeviceContext:
    def __init__(self, dim, chains, target=abi.TARGET_STD_NORMAL, target_params=None, seed=0x23EF614D,
                 max_depth=10, min_delta=-1000.0, chain_offset=0, metric=abi.METRIC_DIAG, device=0,
                 stream=None, dense_per_chain=False):
        self.D, self.C = int(dim), int(chains)
        self.position_epoch = 0      # bumped by every call that can move the chains: whoever caches a position checks it (api.mcmc_next_step)
        cfg = abi.Config()
        cfg.device, cfg.dim, cfg.chains, cfg.chain_offset = device, self.D, self.C, chain_offset
        cfg.metric, cfg.target, cfg.max_depth, cfg.min_delta, cfg.seed = metric, target, max_depth, min_delta, seed
        cfg.dense_per_chain = int(bool(dense_per_chain))   # one dense M⁻¹ per chain, each adapted from its own draws (the reference's semantics)
        if target_params is not None:
            target_params = np.ascontiguousarray(target_params)
            cfg.target_params = target_params.ctypes.data
            cfg.target_params_bytes = target_params.nbytes
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = abi.lib().dhmc_create(C.byref(cfg), C.byref(self.h))
        if rc != abi.OK:
            self.h = None
            self._raise(rc, "dhmc_create")
        if stream is not None:
            self.set_stream(stream)

    # ---- error mapping ---------------------------------------------------------------------
    def _raise(self, rc, what):
        if rc == abi.ERR_INVALID_ARGUMENT:
            detail = abi.lib().dhmc_last_error(self.h).decode() if self.h else ""
            raise ValueError(f"ArgumentError in {what}" + (f": {detail}" if detail else ""))  # the reference's @argcheck failures
        if rc == abi.ERR_CHAIN_FAILURE:
            st = self.status()
            bad = np.nonzero(st)[0]
            bits = int(np.bitwise_or.reduce(st[bad]))
            msg = next(m for b, m in STATUS_MESSAGES if bits & b)
            raise DynamicHMCError(msg, chains=bad[:16].tolist(), n_failed=int(bad.size), status=st[bad[:16]].tolist())
        if rc == abi.ERR_CALLBACK and getattr(self, "_cb_error", None) is not None:
            err, self._cb_error = self._cb_error, None
            raise err                                # the exception the user's log density raised
        detail = abi.lib().dhmc_last_error(self.h).decode() if self.h else ""
        raise RuntimeError(f"{what}: {abi.ERROR_NAMES.get(rc, rc)} {detail}")

    def _chk(self, rc, what, allow_failure=False):
        if rc == abi.OK or (allow_failure and rc == abi.ERR_CHAIN_FAILURE):
            return rc
        self._raise(rc, what)

    def close(self):
        if getattr(self, "h", None):
            abi.lib().dhmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state -----------------------------------------------------------------------------
    def set_stream(self, stream):
        self._chk(abi.lib().dhmc_set_stream(self.h, C.c_void_p(int(stream))), "dhmc_set_stream")

    def init(self, q0=None, allow_failure=False):
        """initialize_warmup_state (mcmc.jl:129-132)."""
        if q0 is not None and isinstance(q0, np.ndarray):
            q0 = np.ascontiguousarray(np.broadcast_to(q0, (self.C, self.D)), np.float64)
        self.position_epoch += 1
        return self._chk(abi.lib().dhmc_init(self.h, _ptr(q0), int(q0 is not None and _is_device(q0))), "dhmc_init", allow_failure)

    def set_position(self, q, allow_failure=False):
        """Q := evaluate_ℓ(ℓ, q) at the caller's positions; κ, ϵ, adaptation state and random streams are kept."""
        if isinstance(q, np.ndarray) or not _is_device(q):
            q = np.ascontiguousarray(np.broadcast_to(np.asarray(q, np.float64), (self.C, self.D)), np.float64)
        self.position_epoch += 1
        return self._chk(abi.lib().dhmc_set_position(self.h, _ptr(q), int(_is_device(q))), "dhmc_set_position", allow_failure)

    def position(self):
        q = np.zeros((self.C, self.D)); lq = np.zeros(self.C); g = np.zeros((self.C, self.D))
        self._chk(abi.lib().dhmc_get_position(self.h, _ptr(q), _ptr(lq), _ptr(g), 0), "dhmc_get_position")
        return q, lq, g

    def set_metric_diag(self, minv):
        """GaussianKineticEnergy(Diagonal(minv)) (hamiltonian.jl:80): [D] shared, or [C][D] per chain (numpy, or a CUDA tensor)."""
        if _is_device(minv):
            minv = minv.contiguous()
            self._chk(abi.lib().dhmc_set_metric_diag(self.h, _ptr(minv), int(minv.dim() == 2), 1), "dhmc_set_metric_diag")
            return
        minv = np.ascontiguousarray(minv, np.float64)
        self._chk(abi.lib().dhmc_set_metric_diag(self.h, _ptr(minv), int(minv.ndim == 2), 0), "dhmc_set_metric_diag")

    def metric_diag(self):
        m = np.zeros((self.C, self.D))
        self._chk(abi.lib().dhmc_get_metric_diag(self.h, _ptr(m), 0), "dhmc_get_metric_diag")
        return m

    def set_metric_dense(self, minv):
        """GaussianKineticEnergy(M⁻¹) with a full matrix shared by all chains (hamiltonian.jl:73)."""
        if _is_device(minv):
            minv = minv.contiguous()
            self._chk(abi.lib().dhmc_set_metric_dense(self.h, _ptr(minv), 1), "dhmc_set_metric_dense")
            return
        minv = np.ascontiguousarray(minv, np.float64)
        self._chk(abi.lib().dhmc_set_metric_dense(self.h, _ptr(minv), 0), "dhmc_set_metric_dense")

    def set_dense_products(self, products):
        """1: one M⁻¹ product per leapfrog (the default of a shared dense metric); 2: the reference's two (include/dhmc.h)."""
        self._chk(abi.lib().dhmc_set_dense_products(self.h, int(products)), "dhmc_set_dense_products")

    def dense_products(self):
        return int(abi.lib().dhmc_get_dense_products(self.h))

    def metric_dense(self, chain=0):
        """(M⁻¹, W) of the shared dense metric, or of `chain` in a dense_per_chain context."""
        m = np.zeros((self.D, self.D)); W = np.zeros((self.D, self.D))
        self._chk(abi.lib().dhmc_get_metric_dense_chain(self.h, C.c_int32(chain), _ptr(m), _ptr(W)), "dhmc_get_metric_dense_chain")
        return m, W

    def set_stepsize(self, eps):
        eps = np.ascontiguousarray(np.atleast_1d(eps), np.float64)
        if eps.size not in (1, self.C):
            raise ValueError(f"ArgumentError: ϵ must be a scalar or one value per chain ({self.C}), got {eps.size}")
        self._chk(abi.lib().dhmc_set_stepsize(self.h, _ptr(eps), int(eps.size == self.C), 0), "dhmc_set_stepsize")

    def stepsize(self):
        e = np.zeros(self.C)
        self._chk(abi.lib().dhmc_get_stepsize(self.h, _ptr(e), 0), "dhmc_get_stepsize")
        return e

    def status(self):
        s = np.zeros(self.C, np.uint32)
        rc = abi.lib().dhmc_get_status(self.h, _ptr(s))
        if rc != abi.OK:
            raise RuntimeError("dhmc_get_status failed")
        return s

    def find_initial_stepsize(self, initial_eps=0.1, log_threshold=float(np.log(0.8)), maxiter_crossing=400,
                              allow_failure=False):
        p = abi.StepsizeSearch(initial_eps, log_threshold, maxiter_crossing, 0)
        return self._chk(abi.lib().dhmc_find_initial_stepsize(self.h, C.byref(p)), "dhmc_find_initial_stepsize", allow_failure)

    # ---- the per-draw loops ----------------------------------------------------------------
    def run_into(self, N, arrays, da=None, allow_failure=False):
        """dhmc_run into caller-owned buffers (`arrays`: name -> numpy array or CUDA torch tensor)."""
        o = abi.Outputs()
        dev = [_is_device(a) for a in arrays.values() if a is not None]
        if dev and any(dev) != all(dev):
            raise ValueError("output buffers must be all host or all device")
        o.on_device = int(bool(dev) and all(dev))
        for name, _ in abi.OUTPUT_FIELDS:
            a = arrays.get(name)
            setattr(o, name, _ptr(a).value if a is not None else None)
        dap = None
        if da is not None:
            d = dict(delta=0.8, gamma=0.05, kappa=0.75, t0=10, init=1, finalize=1)
            d.update(da)
            dap = abi.DualAveragingABI(d["delta"], d["gamma"], d["kappa"], d["t0"], d["init"], d["finalize"], 0)
        self.position_epoch += 1
        rc = abi.lib().dhmc_run(self.h, C.c_int64(N), C.byref(dap) if dap is not None else None, C.byref(o))
        return self._chk(rc, "dhmc_run", allow_failure)

    def run(self, N, da=None, fields=None, allow_failure=False, pinned=True):
        """dhmc_run into fresh numpy arrays.  pinned: the arrays live in page-locked memory (pinned_empty), which the
        library fills at PCIe speed and under the next chunk's kernel; they behave like any other numpy array."""
        arrs = {}
        for name, dt in abi.OUTPUT_FIELDS:
            if fields is not None and name not in fields:
                continue
            shape = (self.C, N, self.D) if name == "draws" else (self.C, N)
            arrs[name] = pinned_empty(shape, dt) if pinned else np.zeros(shape, dt)
        self.run_into(N, arrs, da=da, allow_failure=allow_failure)
        return arrs

    def update_metric_diag(self, draws, lam=0.0):
        """κ := GaussianKineticEnergy(regularize(sample_M⁻¹(Diagonal, draws), λ)) (mcmc.jl:281-284)."""
        if isinstance(draws, np.ndarray):
            draws = np.ascontiguousarray(draws, np.float64)
        n = draws.shape[1]
        self._chk(abi.lib().dhmc_update_metric_diag(self.h, _ptr(draws), C.c_int64(n), C.c_double(lam), int(_is_device(draws))),
                  "dhmc_update_metric_diag")

    def metric_window_begin(self):
        """Opens a metric window (include/dhmc.h dhmc_metric_window_begin): from now on every transition's draw joins per-chain running
        moments on the device, so a tuning stage needs no [C][N][D] posterior matrix for its metric update."""
        self._chk(abi.lib().dhmc_metric_window_begin(self.h), "dhmc_metric_window_begin")

    def metric_window_count(self):
        return int(abi.lib().dhmc_metric_window_count(self.h))

    def metric_window_end(self):
        self._chk(abi.lib().dhmc_metric_window_end(self.h), "dhmc_metric_window_end")

    def update_metric_diag_window(self, lam=0.0):
        """κ := GaussianKineticEnergy(Diagonal(window variance)) per chain from the open window's moments; closes the window."""
        self._chk(abi.lib().dhmc_update_metric_diag_window(self.h, C.c_double(lam)), "dhmc_update_metric_diag_window")

    def update_metric_dense(self, draws, lam):
        """Pooled dense estimate: κ := GaussianKineticEnergy(regularize(Symmetric(cov(draws)), λ)) (mcmc.jl:210,218-222)."""
        if isinstance(draws, np.ndarray):
            draws = np.ascontiguousarray(draws, np.float64)
        self._chk(abi.lib().dhmc_update_metric_dense(self.h, _ptr(draws), C.c_int64(draws.shape[1]), C.c_double(lam), int(_is_device(draws))),
                  "dhmc_update_metric_dense")

    def set_metric_allreduce(self, allreduce):
        """The shared dense metric adapted from the draws of ALL ranks (include/dhmc.h dhmc_set_metric_allreduce): `allreduce(t)`
        adds the CUDA float64 tensor `t` over the ranks in place (sharding.TorchAllReduce wraps torch.distributed); None: back
        to pooling over this context's chains only."""
        if allreduce is None:
            self._ar = None
            return self._chk(abi.lib().dhmc_set_metric_allreduce(self.h, abi.ALLREDUCE_FN(0), None), "dhmc_set_metric_allreduce")
        import torch

        class _Dev:
            def __init__(s, ptr, n):
                s.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(int(ptr), False), version=2, strides=(8,))
        dev = torch.device("cuda", self.cfg.device)
        self._cb_error = None

        def trampoline(user, ptr, count, stream):
            try:
                ctxm = torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev))
                with ctxm:
                    allreduce(torch.as_tensor(_Dev(ptr, int(count)), device=dev))
                return 0
            except Exception as e:          # no exception may cross the C ABI
                self._cb_error = e
                return 1
        self._ar = abi.ALLREDUCE_FN(trampoline)
        return self._chk(abi.lib().dhmc_set_metric_allreduce(self.h, self._ar, None), "dhmc_set_metric_allreduce")

    # ---- DHMC_TARGET_EXTERNAL: the user's own batched log density --------------------------------
    def set_logdensity_callback(self, fn):
        """`fn(q) -> (lq, grad)` with q a CUDA torch tensor [C][D] (a view of the library's buffer: do not keep it),
        lq [C], grad [C][D]: the LogDensityProblems.logdensity_and_gradient of all chains at once (hamiltonian.jl:204).
        Called once per leapfrog round on the context's stream."""
        import torch

        class _Dev:     # a raw device pointer as a zero-copy torch tensor
            def __init__(s, ptr, shape, strides):
                s.__cuda_array_interface__ = dict(shape=shape, typestr="<f8", data=(int(ptr), False), version=2, strides=strides)
        dev = torch.device("cuda", self.cfg.device)
        self._cb_error = None

        def trampoline(user, q_ptr, chains, ld, dim, lq_ptr, grad_ptr, stream):
            try:
                # run the model on the stream the library's kernels are on (a null handle is the device's default stream)
                ctxm = torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev))
                with ctxm:
                    q = torch.as_tensor(_Dev(q_ptr, (chains, ld), (ld * 8, 8)), device=dev)[:, :dim]
                    lq_out = torch.as_tensor(_Dev(lq_ptr, (chains,), (8,)), device=dev)
                    g_out = torch.as_tensor(_Dev(grad_ptr, (chains, ld), (ld * 8, 8)), device=dev)
                    lq, g = fn(q)
                    lq_out.copy_(lq.to(torch.float64))
                    g_out[:, :dim].copy_(g.to(torch.float64))
                return 0
            except Exception as e:          # no exception may cross the C ABI
                self._cb_error = e
                return 1
        self._cb = abi.LOGDENSITY_FN(trampoline)      # keep the ctypes thunk alive as long as the context
        self._chk(abi.lib().dhmc_set_logdensity_callback(self.h, self._cb, None), "dhmc_set_logdensity_callback")

    # ---- Diagnostics probes (src/diagnostics.jl) ---------------------------------------------
    def _probe_chk(self, rc, what, status, allow_failure):
        if rc == abi.ERR_CHAIN_FAILURE and not allow_failure:
            bad = np.nonzero(status)[0]
            bits = int(np.bitwise_or.reduce(status[bad]))
            msg = next(m for b, m in STATUS_MESSAGES if bits & b)
            raise DynamicHMCError(msg, chains=bad[:16].tolist(), n_failed=int(bad.size), status=status[bad[:16]].tolist())
        return self._chk(rc, what, allow_failure)

    def leapfrog_trajectory(self, eps, first, last, p=None, momentum_index=0, with_points=True, allow_failure=False):
        """leapfrog_trajectory (diagnostics.jl:214-227) for every chain from its current position."""
        npos = int(last) - int(first) + 1
        if npos < 1:
            raise ValueError("ArgumentError in dhmc_leapfrog_trajectory")
        out = dict(delta=np.zeros((self.C, npos)), logdensity=np.zeros((self.C, npos)),
                   range=np.zeros((self.C, 2), np.int32), status=np.zeros(self.C, np.uint32),
                   q=np.zeros((self.C, npos, self.D)) if with_points else None,
                   p=np.zeros((self.C, npos, self.D)) if with_points else None)
        if p is not None:
            p = np.ascontiguousarray(np.broadcast_to(p, (self.C, self.D)), np.float64)
        rc = abi.lib().dhmc_leapfrog_trajectory(self.h, C.c_double(eps), C.c_int32(first), C.c_int32(last),
                                                C.c_uint32(momentum_index), _ptr(p), _ptr(out["delta"]),
                                                _ptr(out["logdensity"]), _ptr(out["q"]), _ptr(out["p"]),
                                                _ptr(out["range"]), _ptr(out["status"]))
        self._probe_chk(rc, "dhmc_leapfrog_trajectory", out["status"], allow_failure)
        return out

    def explore_log_acceptance_ratios(self, eps, n_momenta=20, ps=None, momentum_index=0, allow_failure=False):
        """explore_log_acceptance_ratios (diagnostics.jl:144-152): [C][n_momenta][len(eps)]."""
        eps = np.ascontiguousarray(np.atleast_1d(eps), np.float64)
        if ps is not None:
            ps = np.ascontiguousarray(ps, np.float64)
            if ps.ndim == 2:
                ps = np.ascontiguousarray(np.broadcast_to(ps, (self.C,) + ps.shape))
            n_momenta = ps.shape[1]
        out = np.zeros((self.C, n_momenta, eps.size))
        status = np.zeros(self.C, np.uint32)
        rc = abi.lib().dhmc_explore_log_acceptance_ratios(self.h, _ptr(eps), C.c_int32(eps.size), C.c_int32(n_momenta),
                                                          C.c_uint32(momentum_index), _ptr(ps), _ptr(out), _ptr(status))
        self._probe_chk(rc, "dhmc_explore_log_acceptance_ratios", status, allow_failure)
        return out

    # ---- resume ----------------------------------------------------------------------------
    def export_state(self):
        n = C.c_uint64()
        self._chk(abi.lib().dhmc_state_bytes(self.h, C.byref(n)), "dhmc_state_bytes")
        blob = np.zeros(n.value, np.uint8)
        self._chk(abi.lib().dhmc_export_state(self.h, _ptr(blob), n), "dhmc_export_state")
        return blob

    def import_state(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        self.position_epoch += 1
        self._chk(abi.lib().dhmc_import_state(self.h, _ptr(blob), C.c_uint64(blob.size)), "dhmc_import_state")

    # ---- measurement -----------------------------------------------------------------------
    def last_run_kernel_ms(self):
        return abi.lib().dhmc_last_run_kernel_ms(self.h)

    def last_run_leapfrogs(self):
        return int(abi.lib().dhmc_last_run_leapfrogs(self.h))

    def last_run_rounds(self):
        return int(abi.lib().dhmc_last_run_rounds(self.h))

    def workspace_bytes(self):
        return int(abi.lib().dhmc_workspace_bytes(self.h))
