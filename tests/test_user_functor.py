"""CPU: a caller's device functor compiles against the library's kernel templates with hiprtc (no GPU needed for the compile
step: dhmc_check_target_source); a broken one comes back with the compiler's log.  The run itself: tests/test_gpu_user_functor.py."""
import pytest

import user_functors as uf
from __graft_entry__ import load_package


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.mark.parametrize("dim", [3, 100, 1000, 1500, 4000])
def test_user_functors_compile_into_the_kernels(pkg, dim):
    """(beyond 1024 coordinates only the batched evaluation kernel is compiled: functor_eval_kernel, nuts_kernels.hpp)"""
    for src, name in ((uf.DIAG_NORMAL, "MyDiagNormal"), (uf.STUDENT_T, "StudentT")):
        ok, log = pkg.DeviceFunctorLogDensity.check(dim, src, name)
        assert ok, log


@pytest.mark.parametrize("dim", [100, 1000])
def test_user_functors_compile_into_the_dense_metric_kernels(pkg, dim):
    """Round engine K0 / K2 / K3 (the workgroup-per-chain K3b from 512 coordinates), the wave-per-chain dense kernels, the probes."""
    ok, log = pkg.DeviceFunctorLogDensity.check(dim, uf.STUDENT_T, "StudentT", metric=pkg.abi.METRIC_DENSE)
    assert ok, log


def test_the_references_mixture_case_compiles_as_a_functor(pkg):
    """mix(0.2, N(0, I₃), N(1, L₂L₂ᵀ)) of test/sample-correctness_tests.jl:89-98 as a device functor (log-sum-exp of two quadratic
    forms with the ABI's scalar math); the run and the reference's bars: tests/test_gpu_sample_correctness.py."""
    ok, log = pkg.DeviceFunctorLogDensity.check(3, uf.MIXTURE3, "Mixture3")
    assert ok, log
    ok, log = pkg.DeviceFunctorLogDensity.check(3, uf.MIXTURE3, "Mixture3", metric=pkg.abi.METRIC_DENSE)
    assert ok, log


def test_compile_errors_come_back_with_the_log(pkg):
    ok, log = pkg.DeviceFunctorLogDensity.check(10, uf.BROKEN, "Broken")
    assert not ok
    assert "undeclared_thing" in log
    ok, log = pkg.DeviceFunctorLogDensity.check(10, uf.DIAG_NORMAL, "NoSuchStruct")
    assert not ok and "NoSuchStruct" in log
    ok, _ = pkg.DeviceFunctorLogDensity.check(5000, uf.DIAG_NORMAL, "MyDiagNormal")      # beyond 4096 coordinates: unsupported
    assert not ok
