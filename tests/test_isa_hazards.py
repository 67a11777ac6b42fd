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


def test_matrix_instructions_accumulate_in_vgprs():
    """On gfx950 v_mfma_f64_16x16x4_f64 issues at about half the rate when its accumulators are AGPRs (profiles/r06_mfma_vgpr_form.txt:
    47 against 77 TFLOP/s from registers); csrc/Makefile builds with -mllvm -amdgpu-mfma-vgpr-form.  Every matrix instruction of the built
    library must have an architectural VGPR destination."""
    import collections, re, tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazard_verify as hv
    lib = LIB
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    agpr, total = collections.Counter(), 0
    for obj in hv.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj); f.flush()
            text = subprocess.run([hv.OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        fn = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                fn = m.group(1)
            elif "v_mfma" in line:
                total += 1
                if line.split("v_mfma", 1)[1].split()[1].startswith("a"):
                    agpr[fn] += 1
    assert total > 0, "no matrix instruction found: is this the right library?"
    assert not agpr, "matrix instructions with AGPR accumulators: %s" % dict(agpr)
