"""ctypes harness over tests/hostsim/libpacked_hostsim.so: the CPU simulation of the packed per-draw kernel
(dynamichmc.jl_amd/csrc/packed_body.inc compiled by g++ with one lane per chain).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "packed_hostsim.cpp")
LIB = os.path.join(HERE, "hostsim", "libpacked_hostsim.so")
CSRC = os.path.join(os.path.dirname(HERE), "dynamichmc.jl_amd", "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
DEPS = [SRC] + [os.path.join(CSRC, f) for f in ("packed_body.inc", "packed_core.hpp", "run_params.hpp")] + \
       [os.path.join(INC, f) for f in ("dhmc.h", "dhmc_detmath.h", "dhmc_detmath_tables.h")]

P = C.c_void_p


class ChainArrays(C.Structure):
    _fields_ = [(n, P) for n in ("q", "g", "lq", "minv", "W", "eps", "da", "transition", "status", "ws")]


class DeviceOutputs(C.Structure):
    _fields_ = [(n, P) for n in ("draws", "logdensities", "eps", "pi", "acceptance_rate", "steps", "term_left", "term_right",
                                 "depth", "directions")]


class TargetParams(C.Structure):
    _fields_ = [("a", P), ("b", P), ("c", P), ("n", C.c_int64), ("npad", C.c_int64), ("Dpad", C.c_int32), ("pad_", C.c_int32)]


class RunParams(C.Structure):      # csrc/run_params.hpp, field for field
    _fields_ = [("D", C.c_int), ("Dpad", C.c_int), ("C", C.c_int), ("chain_offset", C.c_int), ("max_depth", C.c_int), ("nvec", C.c_int),
                ("l1_in_lds", C.c_int), ("chain_base", C.c_int), ("k3_block", C.c_int), ("one_product", C.c_int), ("fuse_k2", C.c_int),
                ("min_delta", C.c_double), ("seed", C.c_uint64), ("N", C.c_int64), ("out_stride", C.c_int64),
                ("st", ChainArrays), ("adapt", C.c_int), ("da_init", C.c_int), ("da_finalize", C.c_int), ("t0", C.c_int),
                ("delta", C.c_double), ("gamma", C.c_double), ("kappa", C.c_double), ("out", DeviceOutputs), ("tp", TargetParams),
                ("leapfrog_counter", P), ("win_mean", P), ("win_m2", P), ("win_n0", C.c_int64), ("chain_work", P), ("launch_order", P),
                ("pk_lds_levels", C.c_int), ("pk_align", C.c_int), ("pk_cpl", C.c_int), ("pk_order_base", C.c_int),
                ("pk_queue", C.c_void_p), ("pk_max_waves", C.c_int),
                ("prog", P), ("pk_evicted", P), ("pk_evict_count", P),
                ("pk_live", P), ("pk_handover_below", C.c_int)]


DA_DTYPE = np.dtype([("mu", "f8"), ("Hbar", "f8"), ("logeps", "f8"), ("logeps_bar", "f8"), ("m", "i8")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unknown-pragmas", "-shared",
                            "-o", LIB, SRC], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
        assert _lib.hostsim_sizeof_runparams() == C.sizeof(RunParams), "RunParams layout changed: update tests/hostsim_lib.py"
    return _lib


def _p(a):
    return a.ctypes.data if a is not None else None


class HostSim:
    """The chain state the device context holds ([C][64] padded rows), advanced by the simulated packed kernel."""

    OUT = [("draws", np.float64), ("logdensities", np.float64), ("eps", np.float64), ("pi", np.float64), ("acceptance_rate", np.float64),
           ("steps", np.int64), ("term_left", np.int64), ("term_right", np.int64), ("depth", np.int32), ("directions", np.uint32)]

    def __init__(self, D, chains, target, q, lq, g, eps, minv=None, seed=0x23EF614D, max_depth=10, min_delta=-1000.0, chain_offset=0,
                 params=None, align=4, lds_levels=3):
        assert D <= 64
        self.D, self.C, self.target = D, chains, target
        self.seed, self.max_depth, self.min_delta, self.chain_offset = seed, max_depth, min_delta, chain_offset
        self.align, self.lds_levels = align, lds_levels

        def pad(a, fill=0.0):
            out = np.full((chains, 64), fill)
            out[:, :D] = a
            return out
        self.q, self.g = pad(q), pad(g)
        self.lq = np.array(lq, np.float64)
        self.set_metric(np.ones((chains, D)) if minv is None else minv)
        self.eps = np.broadcast_to(np.asarray(eps, np.float64), (chains,)).copy()
        self.da = np.zeros(chains, DA_DTYPE)
        self.transition = np.zeros(chains, np.uint32)
        self.status = np.zeros(chains, np.uint32)
        self.params = [None, None]
        if params is not None and np.ndim(params[1]) == 2:      # dense-precision normal: mu, the symmetric P as [64][64], zero padded
            mu = np.zeros(64); Pm = np.zeros((64, 64))
            mu[:D] = params[0]
            Pm[:D, :D] = np.triu(params[1]) + np.triu(params[1], 1).T
            self.params = [mu, Pm]
        elif params is not None:            # diagonal normal: mu, prec (padded rows of 64)
            mu = np.zeros(64); prec = np.zeros(64)
            mu[:D], prec[:D] = params[0], params[1]
            self.params = [mu, prec]
        self.win = None
        self.win_n = -1
        self.leapfrogs = np.zeros(1, np.uint64)

    def set_metric(self, minv):
        minv = np.broadcast_to(np.asarray(minv, np.float64), (self.C, self.D))
        self.minv = np.ones((self.C, 64)); self.minv[:, :self.D] = minv
        self.W = np.zeros((self.C, 64)); self.W[:, :self.D] = np.sqrt(1.0 / minv)

    def window_begin(self):
        self.win = np.zeros((2, self.C, 64))
        self.win_n = 0

    def window_update_metric(self):
        var = self.win[1][:, :self.D] / float(self.win_n - 1)
        self.set_metric(var)
        self.win, self.win_n = None, -1

    def _launch(self, out, N, N_total, da, queue, order, prog=None, evicted=None, evict_count=None, chain_work=None,
                leap_total=None, live=None, handover_below=0):
        """One launch of the simulated kernel: the places of `order` (default: every chain) up to transition N of the call."""
        R = RunParams()
        R.D, R.Dpad, R.chain_offset, R.max_depth = self.D, 64, self.chain_offset, self.max_depth
        R.nvec = lib().hostsim_ws_nvec(self.max_depth)
        R.min_delta, R.seed, R.N, R.out_stride = self.min_delta, self.seed, N, N_total
        for k, a in (("q", self.q), ("g", self.g), ("lq", self.lq), ("minv", self.minv), ("W", self.W), ("eps", self.eps), ("da", self.da),
                     ("transition", self.transition), ("status", self.status)):
            setattr(R.st, k, _p(a))
        if da is not None:
            d = dict(delta=0.8, gamma=0.05, kappa=0.75, t0=10, init=1, finalize=1)
            d.update(da)
            R.adapt, R.da_init, R.da_finalize, R.t0 = 1, d["init"], d["finalize"], d["t0"]
            R.delta, R.gamma, R.kappa = d["delta"], d["gamma"], d["kappa"]
        for k, _ in self.OUT:
            setattr(R.out, k, _p(out[k]))
        R.tp.a, R.tp.b, R.tp.Dpad = _p(self.params[0]), _p(self.params[1]), 64
        R.leapfrog_counter = _p(self.leapfrogs)
        if self.win is not None:
            R.win_mean, R.win_m2, R.win_n0 = _p(self.win[0]), _p(self.win[1]), self.win_n
        R.pk_lds_levels, R.pk_align = self.lds_levels, self.align
        R.C = self.C
        if order is not None:
            order = np.ascontiguousarray(order, np.int32)
            R.launch_order = _p(order)
            R.C = len(order)                 # the launch's places
        if prog is not None:
            R.prog = _p(prog)
            R.pk_evicted, R.pk_evict_count, R.chain_work = _p(evicted), _p(evict_count), _p(chain_work)
            R.pk_live, R.pk_handover_below = _p(live), handover_below
        rc = lib().hostsim_packed_run(self.target, C.byref(R), int(queue), self.C)
        assert rc == 0, rc

    def _outputs(self, N):
        return {k: np.zeros((self.C, N, self.D) if k == "draws" else (self.C, N), t) for k, t in self.OUT}

    def run(self, N, da=None, queue=False, order=None):
        """One launch of N transitions per chain.  queue: through the kernel's queue of places (one lane group walks every chain);
        order: the launch order (a permutation of the chains)."""
        out = self._outputs(N)
        self.leapfrogs[:] = 0
        self._launch(out, N, 0, da, queue, order)
        if self.win is not None:
            self.win_n += N
        return out

    def run_handover(self, N, da=None):
        """The end game of a packed launch (RunParams::pk_live, pk_handover_below) at its extreme: one lane group, hand-over threshold
        one — every launch gives a chain up after one transition, and the call is N launches over the chains that are left (the
        device hands the chains it gives up to the pipeline kernel).  Returns the outputs and the number of launches."""
        out = self._outputs(N)
        self.leapfrogs[:] = 0
        prog = np.zeros(self.C, np.int32)
        work = np.zeros(self.C, np.uint32)
        evicted, count, live = np.zeros(self.C, np.int32), np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        left = np.arange(self.C, dtype=np.int32)[::-1].copy()
        launches = 0
        while len(left):
            count[:] = 0
            live[:] = 1
            self._launch(out, N, N, da, True, left, prog, evicted, count, work, live=live, handover_below=1)
            left = np.sort(evicted[:int(count[0])]).astype(np.int32)
            launches += 1
            assert launches <= N + 1
        assert (prog == N).all() and int(work.sum()) == int(self.leapfrogs[0]) == int(out["steps"].sum())
        if self.win is not None:
            self.win_n += N
        return out, launches
