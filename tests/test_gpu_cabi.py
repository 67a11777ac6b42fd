"""The C ABI from a plain-C client (tests/cabi/cabi_client.c: what a cgo / ccall / JNI binding sees): the header is
valid C99, the program links against libdhmc_amd.so alone, and on the GPU it reproduces the ctypes path's bits."""
import os
import subprocess

import numpy as np
import pytest

from __graft_entry__ import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "cabi_client.c")
LIBDIR = os.path.join(ROOT, "dynamichmc.jl_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "cabi_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-o", exe, SRC, "-I", os.path.join(ROOT, "include"),
                    "-L", LIBDIR, "-ldhmc_amd", f"-Wl,-rpath,{LIBDIR}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_plain_c_client_compiles_and_links(tmp_path):
    load_package()            # the library must exist (built by __graft_entry__.build())
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_plain_c_client_reproduces_ctypes_path(tmp_path):
    pkg = load_package()
    out = subprocess.run([_build(tmp_path)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split(" ", 1) for line in out.strip().splitlines())
    ctx = pkg.DeviceContext(100, 8)
    ctx.init(); ctx.find_initial_stepsize()
    a = ctx.run(30, da={}, fields=["draws", "steps"])
    ctx.update_metric_diag(a["draws"])
    b = ctx.run(30, fields=["draws", "steps"])
    assert float(got["checksum"]) == float(np.cumsum(b["draws"].ravel())[-1])      # the C loop's left-to-right sum
    assert int(got["steps"]) == int(b["steps"].sum())
    assert float(got["eps0"]) == ctx.stepsize()[0]
    assert int(got["leapfrogs"]) == ctx.last_run_leapfrogs()
    ctx.metric_window_begin(); ctx.run_into(30, {}, da={}); ctx.update_metric_diag_window()
    c = ctx.run(30, fields=["draws"])
    assert float(got["checksum_after_window"]) == float(np.cumsum(c["draws"].ravel())[-1])
