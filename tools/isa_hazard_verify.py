"""Verifies the gfx950 data hazards that software must cover with wait states, over the DISASSEMBLY of every kernel in the built
library (or any .so / .o / code object given):

    python tools/isa_hazard_verify.py [dynamichmc.jl_amd/lib/libdhmc_amd.so ...]

Why: the kernels contain inline-asm blocks (Horner chains with scalar-register coefficients, Philox products:
csrc/detmath_dev.hpp, csrc/philox_dev.hpp), and the compiler's hazard recognizer covers its own instructions only.  Round 4's
first GPU run read a stale register exactly that way (an asm v_readfirstlane right behind the v_cvt that produced its input; an
asm v_fma_f64 right behind the v_readlane that reloaded its coefficient from an SGPR spill slot).  Rules checked (LLVM
GCNHazardRecognizer, gfx940 family; a wait state = one issued instruction, `s_nop n` = n + 1):

  1. VALU writes an SGPR / VCC  ->  VALU reads it as an operand (incl. v_cndmask's vcc, carry-ins):  2 wait states
  2. VALU writes a VGPR         ->  v_readlane / v_readfirstlane reads it:                           1 wait state
  3. VALU writes an SGPR / VCC  ->  v_readlane / v_writelane uses it as the lane select:             4 wait states
  4. VALU writes VCC            ->  v_div_fmas reads it implicitly:                                  4 wait states
  5. VALU writes a VGPR         ->  a DPP instruction reads it as its moved operand:                 2 wait states
  6. a transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) writes a VGPR  ->  another VALU reads it:  1 wait state

The walk is along the layout (fall-through order): a conditional branch is an issued instruction like any other, an
unconditional one ends the window (taken edges are the compiler's business; the asm blocks open with their own wait states).  Exit status 1 if any
rule is violated.  `--self-test` feeds it known-bad snippets."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TWO_DST = ("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32",
           "v_subbrev_co_u32", "v_div_scale_f64", "v_div_scale_f32")


def regs(op, kind):
    """register numbers of class `kind` ('s' or 'v') named in one operand; VCC is ('s', 106..107)"""
    out = set()
    op = op.strip().lstrip("-|").rstrip("|")
    if kind == "s":
        if op.startswith("vcc"):
            if op in ("vcc", "vcc_lo"):
                out.add(106)
            if op in ("vcc", "vcc_hi"):
                out.add(107)
            return out
    m = re.fullmatch(r"%s\[(\d+):(\d+)\]" % kind, op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"%s(\d+)" % kind, op)
    if m:
        return {int(m.group(1))}
    return out


def split_ops(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])]
    # drop modifiers that follow the last operand ("v1 quad_perm:[..] row_mask:0xf")
    ops = [o.split()[0] if o and not o.startswith("|") else o for o in ops]
    return parts[0], ops


def analyse(inst):
    """-> (is_valu, sgpr_writes, vgpr_writes, sgpr_reads, vgpr_src0_reads_if_readlane, lane_select_sgprs, reads_vcc_implicitly, wait)"""
    op, ops = split_ops(inst)
    m = re.fullmatch(r"s_nop (\d+)", inst.strip())
    wait = int(m.group(1)) + 1 if m else 1
    if not op.startswith("v_") or op.startswith("v_accvgpr") and False:
        return dict(valu=False, sw=set(), vw=set(), sr=set(), rl=set(), ls=set(), fmas=False, wait=wait, vr=set(), dpp_src=set(), trans=False)
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    ndst = 2 if base in TWO_DST else 1
    if base.startswith(("v_cmpx",)):
        ndst = 1
    dsts, srcs = ops[:ndst], ops[ndst:]
    if base.startswith("v_cmp") and not op.endswith("_e64") and dsts and not dsts[0].startswith(("vcc", "s")):
        dsts, srcs = ["vcc"], ops             # e32 compare written without an explicit vcc
    sw, vw = set(), set()
    for d in dsts:
        sw |= regs(d, "s")
        vw |= regs(d, "v")
    sr = set()
    for s_ in srcs:
        sr |= regs(s_, "s")
    rl, ls = set(), set()
    if base in ("v_readlane_b32", "v_readfirstlane_b32"):
        rl = regs(srcs[0], "v") if srcs else set()
        if base == "v_readlane_b32" and len(srcs) > 1:
            ls = regs(srcs[1], "s")
    if base == "v_writelane_b32" and len(srcs) > 1:
        ls = regs(srcs[1], "s")
    vr = set()
    for s_ in srcs:
        vr |= regs(s_, "v")
    is_dpp = op.endswith("_dpp") or "quad_perm" in inst or "row_" in inst or "wave_" in inst
    dpp_src = regs(srcs[0], "v") if (is_dpp and srcs) else set()
    trans = bool(re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", base)) or base.startswith("v_rcp_iflag")
    return dict(valu=True, sw=sw, vw=vw, sr=sr, rl=rl, ls=ls, fmas=base.startswith("v_div_fmas"), wait=wait, vr=vr, dpp_src=dpp_src, trans=trans)


def check_function(name, insts, report):
    bad = 0
    hist = []   # recent (analysis, text), newest last
    for text in insts:
        a = analyse(text)
        if a["valu"]:
            dist = 0
            for pa, pt in reversed(hist):
                if dist >= 4:
                    break
                if pa["valu"]:
                    why = None
                    if dist < 2 and pa["sw"] & a["sr"]:
                        why = "rule 1 (VALU-written s%s read by VALU after %d wait state(s), 2 needed)" % (sorted(pa["sw"] & a["sr"]), dist)
                    elif dist < 1 and pa["vw"] & a["rl"]:
                        why = "rule 2 (VALU-written v%s read by readlane after %d wait state(s), 1 needed)" % (sorted(pa["vw"] & a["rl"]), dist)
                    elif dist < 4 and pa["sw"] & a["ls"]:
                        why = "rule 3 (VALU-written s%s used as lane select after %d wait state(s), 4 needed)" % (sorted(pa["sw"] & a["ls"]), dist)
                    elif dist < 4 and a["fmas"] and pa["sw"] & {106, 107}:
                        why = "rule 4 (VALU-written vcc read by v_div_fmas after %d wait state(s), 4 needed)" % dist
                    elif dist < 2 and pa["vw"] & a["dpp_src"]:
                        why = "rule 5 (VALU-written v%s read by a DPP instruction after %d wait state(s), 2 needed)" % (sorted(pa["vw"] & a["dpp_src"]), dist)
                    elif dist < 1 and pa["trans"] and not a["trans"] and pa["vw"] & a["vr"]:
                        why = "rule 6 (result v%s of a transcendental read by a VALU instruction after %d wait state(s), 1 needed)" % (sorted(pa["vw"] & a["vr"]), dist)
                    if why:
                        bad += 1
                        report("%s: %s\n      producer: %s\n      consumer: %s" % (name, why, pt, text))
                        break
                dist += pa["wait"]
        hist.append((a, text))
        if len(hist) > 8:
            hist.pop(0)
        if text.split()[0] in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            hist = []       # no fall-through: what follows in the layout is reached from elsewhere
    return bad


def code_objects(path):
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and struct.unpack_from("<H", data, 18)[0] == 224:     # EM_AMDGPU: already a code object
        return [data]
    out, pos = [], 0
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        num = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(num):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx" in triple and sz:
                out.append(data[i + o:i + o + sz])
        pos = i + 24
    return out


def verify(path, report=print):
    nfun = ninst = bad = 0
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as fh:
            fh.write(blob)
            fh.flush()
            txt = subprocess.run([OBJDUMP, "-d", fh.name], capture_output=True, text=True, check=True).stdout
        name, insts = None, []
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if name and insts:
                    bad += check_function(name, insts, report); nfun += 1; ninst += len(insts)
                name, insts = m.group(1), []
                continue
            if line.startswith("\t") and name:
                t = line.split("//")[0].strip()
                if t:
                    insts.append(t)
        if name and insts:
            bad += check_function(name, insts, report); nfun += 1; ninst += len(insts)
    return nfun, ninst, bad


def self_test():
    cases = [
        (["v_readlane_b32 s5, v255, 17", "v_fma_f64 v[0:1], v[0:1], v[2:3], s[4:5]"], 1),
        (["v_readlane_b32 s5, v255, 17", "s_nop 1", "v_fma_f64 v[0:1], v[0:1], v[2:3], s[4:5]"], 0),
        (["v_readlane_b32 s5, v255, 17", "s_nop 0", "v_fma_f64 v[0:1], v[0:1], v[2:3], s[4:5]"], 1),
        (["v_cvt_i32_f64_e32 v1, v[8:9]", "v_readfirstlane_b32 s0, v1"], 1),
        (["v_cvt_i32_f64_e32 v1, v[8:9]", "v_mov_b32_e32 v2, 0", "v_readfirstlane_b32 s0, v1"], 0),
        (["v_cmp_lt_f64_e32 vcc, s[0:1], v[2:3]", "v_cndmask_b32_e32 v5, 0, v1, vcc"], 1),
        (["v_cmp_lt_f64_e32 vcc, s[0:1], v[2:3]", "s_nop 1", "v_cndmask_b32_e32 v5, 0, v1, vcc"], 0),
        (["v_readfirstlane_b32 s3, v1", "s_nop 1", "v_readlane_b32 s4, v2, s3"], 1),
        (["v_mad_u64_u32 v[0:1], vcc, s5, v2, 0", "v_mad_u64_u32 v[4:5], vcc, s6, v3, 0"], 0),
        (["s_mov_b32 s5, 3", "v_fma_f64 v[0:1], v[0:1], v[2:3], s[4:5]"], 0),
    ]
    ok = True
    for insts, want in cases:
        got = check_function("t", insts, lambda *_: None)
        if (got > 0) != (want > 0):
            ok = False
            print("self-test FAILED:", insts, "expected", want, "got", got)
    print("self-test", "ok" if ok else "FAILED")
    return ok


if __name__ == "__main__":
    if "--self-test" in sys.argv:
        sys.exit(0 if self_test() else 1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = [a for a in sys.argv[1:] if not a.startswith("-")] or [os.path.join(root, "dynamichmc.jl_amd", "lib", "libdhmc_amd.so")]
    total = 0
    for p in paths:
        nfun, ninst, bad = verify(p)
        print("%s: %d kernels / functions, %d instructions, %d hazard violation(s)" % (p, nfun, ninst, bad))
        total += bad
    sys.exit(1 if total else 0)
