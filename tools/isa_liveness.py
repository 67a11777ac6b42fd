"""VGPR liveness of a compiled kernel, from its final assembly (no GPU needed).

    python tools/isa_liveness.py file.s [kernel-name-substring]

Standard backward dataflow over the basic blocks of the kernel: which vector registers are live at every instruction.
Prints the peak, a profile of the live count along the layout (with the source lines of `.loc` directives when the file
was compiled with -gline-tables-only), and for the spill instructions where they sit.  Used to find out WHICH part of
nuts_run_kernel decides its register allocation (tools/isa_regions.py gives the instruction mix).
"""
import collections
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.append((m.group(3), i))
    return out


STORE = ("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "ds_add", "global_atomic", "ds_max", "ds_min")
RMW = ("v_fmac", "v_mac", "v_writelane", "v_dot", "v_mfma", "v_pk_fmac")
NOVDEF = ("v_cmp", "v_readlane", "v_readfirstlane", "v_cmpx")
TWODEF = ("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co")


def def_use(op, operands):
    """operands: list of operand strings. returns (defs, uses) as lists of (file, idx)."""
    if not operands:
        return [], []
    if op.startswith(STORE) or op.startswith("s_") or op.startswith("buffer_wbl2") or op.startswith("ds_bpermute") is None:
        uses = []
        for t in operands:
            uses += regs_of(t)
        return [], uses
    if op.startswith(NOVDEF):
        uses = []
        for t in operands[1:]:
            uses += regs_of(t)
        return [], uses
    defs = regs_of(operands[0])
    uses = []
    rest = operands[1:]
    for t in rest:
        uses += regs_of(t)
    if op.startswith(RMW) or "dpp" in op or any("row_" in t or "quad_perm" in t for t in operands):
        uses += defs
    if op.startswith("v_cndmask"):
        pass
    return defs, uses


def parse(path, pat):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and "nuts_run_kernel" in m.group(1) and pat in m.group(1):
            start = i
            break
    if start is None:
        for i, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if m and pat in m.group(1):
                start = i
                break
    insts = []   # (label_before, op, operands, loc, raw)
    labels = {}
    loc = None
    files = {}
    for l in lines[:start]:
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", t)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not l.startswith("\t") or t.startswith(".") or t.startswith(";") or not t:
            continue
        code = t.split(";")[0].strip()
        parts = code.split(None, 1)
        op = parts[0]
        operands = [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []
        insts.append((op, operands, loc, t))
    return insts, labels


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    insts, labels = parse(path, pat)
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (op, operands, loc, raw) in enumerate(insts):
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i].append(labels[operands[0]])
            continue
        if op.startswith("s_cbranch"):
            succ[i].append(labels[operands[0]])
        if op == "s_setpc_b64":
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
    du = [def_use(op, operands) for (op, operands, loc, raw) in insts]
    live_in = [frozenset()] * n
    changed = True
    it = 0
    while changed:
        changed = False
        it += 1
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            d, u = du[i]
            new = (out - set(d)) | set(u)
            if new != live_in[i]:
                live_in[i] = frozenset(new)
                changed = True
    counts = [len([r for r in live_in[i] if r[0] == "v"]) for i in range(n)]
    acounts = [len([r for r in live_in[i] if r[0] == "a"]) for i in range(n)]
    peak = max(counts)
    pi = counts.index(peak)
    print("instructions %d, dataflow passes %d, peak live VGPRs %d (AGPRs %d) at #%d %s %s" % (n, it, peak, max(acounts), pi, insts[pi][2], insts[pi][3][:60]))
    # profile: by source line, the max live count
    byloc = collections.OrderedDict()
    for i in range(n):
        k = insts[i][2]
        if k not in byloc:
            byloc[k] = [0, 0]
        byloc[k][0] = max(byloc[k][0], counts[i])
        byloc[k][1] += 1
    print("-- max live VGPRs by source line (top 40)")
    for k, v in sorted(byloc.items(), key=lambda kv: -kv[1][0])[:40]:
        print("   %-28s live %3d   (%d instructions)" % (k, v[0], v[1]))
    print("-- layout profile (every ~1/60 of the kernel): index, live, source")
    step = max(1, n // 60)
    for i in range(0, n, step):
        j = max(range(i, min(n, i + step)), key=lambda x: counts[x])
        print("   #%5d live %3d  %s" % (j, counts[j], insts[j][2]))
    # registers live at the peak: how long are their ranges (number of instructions they are live at)
    span = collections.Counter()
    for i in range(n):
        for r in live_in[i]:
            span[r] += 1
    at_peak = sorted(live_in[pi], key=lambda r: -span[r])
    print("-- registers live at the peak, by how many instructions they are live at")
    print("   " + " ".join("%s%d:%d" % (r[0], r[1], span[r]) for r in at_peak))
    sp = [(i, insts[i]) for i in range(n) if insts[i][0].startswith("scratch_")]
    print("-- %d scratch instructions" % len(sp))


if __name__ == "__main__":
    main()
