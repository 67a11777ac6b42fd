"""Static check of the software prefetches of the per-draw kernel (wave.hpp prefetch_row, nuts_kernels.hpp lae_prefetch) in a
compiled kernel's assembly:

    python tools/isa_prefetch_verify.py file.s

A prefetch is a load whose result nobody reads; it is issued by an asm statement, so the compiler does not know that the
destination register is written LATER, when the data arrives.  That is safe only if nothing else lives in that register while a
prefetch can be in flight:
  * vector prefetches ("; dhmc_pf") all write ONE register (the PrefetchToken: an accumulation register, alive for the whole kernel), and no other instruction
    of the kernel writes that register (no live-range split, no spill-and-reload of the token);
  * scalar prefetches ("; dhmc_pf_s") write the SGPRs that the retiring s_waitcnt ("; dhmc_pf_s_retire sA sB") names.
Exit status 1 on a violation."""
import re
import sys


def dest_regs(text):
    """VGPR numbers written by an instruction (first operand; two for the two-destination forms)"""
    parts = text.split(None, 1)
    if len(parts) < 2 or not parts[0].startswith(("v_", "global_load", "buffer_load", "flat_load", "ds_read", "scratch_load")):
        return set()
    if parts[0].startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    op = parts[1].split(",")[0].strip()
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", op)
    if m:
        return {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va]\d+)", op)
    return {m.group(1)} if m else set()


def check(path):
    bad = 0
    kernels = {}
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur and line.startswith("\t"):
            kernels[cur].append(line.strip())
    for name, lines in kernels.items():
        toks = set()
        sload, sretire = set(), set()
        for t in lines:
            if "; dhmc_pf_s_retire" in t:
                sretire |= set(re.findall(r"\bs(\d+)\b", t.split("dhmc_pf_s_retire")[1]))
            elif "; dhmc_pf_s" in t:
                sload.add(re.match(r"s_load_dword s(\d+)", t).group(1))
            elif "; dhmc_pf_init" in t:
                toks |= set(re.findall(r"\b([va]\d+)\b", t.split("dhmc_pf_init")[1]))
            elif re.search(r"; dhmc_pf$", t):
                toks |= dest_regs(t.split(";")[0])
            elif "; dhmc_pf_keep" in t:
                toks |= set(re.findall(r"\b([va]\d+)\b", t.split("dhmc_pf_keep")[1]))
        if not toks and not sload:
            continue
        if len(toks) > 1:
            print("%s: the prefetch token lives in several registers: %s" % (name, sorted(toks)))
            bad += 1
        for t in lines:
            if "dhmc_pf" in t:
                continue
            code = t.split(";")[0]
            if dest_regs(code) & toks:
                print("%s: another instruction writes the token register: %s" % (name, t))
                bad += 1
        # between a scalar prefetch and the next retire in layout order (the merges' vector blocks): nothing else writes the tokens
        open_tok = set()
        for t in lines:
            if "; dhmc_pf_s_retire" in t:
                open_tok = set()
            elif "; dhmc_pf_s" in t:
                open_tok.add(int(re.match(r"s_load_dword s(\d+)", t).group(1)))
            elif open_tok:
                code = t.split(";")[0]
                parts = code.split(None, 1)
                if len(parts) == 2 and parts[0].startswith(("s_", "v_readlane", "v_readfirstlane", "v_cmp")) and not parts[0].startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_cmp", "s_bitcmp")):
                    d = parts[1].split(",")[0].strip()
                    m = re.fullmatch(r"s\[(\d+):(\d+)\]", d)
                    w = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else ({int(d[1:])} if re.fullmatch(r"s\d+", d) else set())
                    if w & open_tok:
                        print("%s: a scalar prefetch token is overwritten while in flight: %s" % (name, t))
                        bad += 1
        if sload != sretire:
            print("%s: scalar prefetches write s%s, retired as s%s" % (name, sorted(sload), sorted(sretire)))
            bad += 1
        print("%s: vector token %s, scalar tokens s%s: %s" % (name[:60], sorted(toks), sorted(sload), "ok" if not bad else "VIOLATION"))
    return bad


if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1]) else 0)
