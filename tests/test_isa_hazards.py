"""CPU: the built HIP library has no uncovered gfx950 data hazard (tools/isa_hazard_verify.py).  The kernels carry empty
inline-asm statements (register-class pins) and, in round 4's first version, carried asm instructions whose hazards the
compiler does not track; a stale scalar operand shows up as a chain that silently leaves the oracle, so the rules the
compiler applies to its own code are re-checked here over the disassembly of everything that ships."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "isa_hazard_verify.py")
LIB = os.path.join(ROOT, "dynamichmc.jl_amd", "lib", "libdhmc_amd.so")


def test_verifier_flags_known_bad_sequences():
    assert subprocess.run([sys.executable, TOOL, "--self-test"], capture_output=True, text=True).returncode == 0


def test_built_library_has_no_hazard_violation():
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    r = subprocess.run([sys.executable, TOOL, LIB], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 hazard violation(s)" in r.stdout and "kernels / functions" in r.stdout
