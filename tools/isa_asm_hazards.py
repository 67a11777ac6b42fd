"""Static check of the inline-asm blocks in a compiled kernel (gfx950): the compiler tracks the `VALU writes SGPR -> VALU reads
it` (2 wait states) and `VALU writes VGPR -> v_readlane reads it` (1) hazards for its own instructions only, so an asm block
that reads an SGPR must not sit within two instructions of a VALU write of that SGPR (v_readlane / v_readfirstlane — SGPR spill
reloads are exactly that — or a VOP3 compare).  Lists every asm block whose scalar inputs were VALU-written in the window before
it, counting the wait states the block itself opens with (s_nop n).

    python tools/isa_asm_hazards.py file.s [window=3]
"""
import re
import sys


def sregs(text):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    if "vcc" in text:
        out.add(-1)
    return out


def main():
    path = sys.argv[1]
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lines = [l.rstrip("\n") for l in open(path)]
    insts = []          # (index in file, text, in_asm_block_id or None)
    blk = None
    nblk = 0
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            blk = nblk; nblk += 1; continue
        if t.startswith(";;#ASMEND"):
            blk = None; continue
        if not l.startswith("\t") or not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^[\w.$]+:", t):
                insts.append((i, "LABEL", None))
            continue
        insts.append((i, t, blk))
    bad = 0
    checked = 0
    k = 0
    while k < len(insts):
        if insts[k][2] is None:
            k += 1; continue
        b = insts[k][2]
        j = k
        while j < len(insts) and insts[j][2] == b:
            j += 1
        body = [insts[x][1] for x in range(k, j)]
        own_wait = 0
        first = body[0]
        m = re.match(r"s_nop (\d+)", first)
        if m:
            own_wait = int(m.group(1)) + 1
        reads = set()
        for t in body:
            ops = t.split(None, 1)[1] if " " in t else ""
            parts = [p.strip() for p in ops.split(",")]
            ndst = 2 if t.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_div_scale")) else 1   # vdst, carry-out
            for p in parts[ndst:]:
                reads |= sregs(p)
        if reads:
            checked += 1
            dist = own_wait
            x = k - 1
            while x >= 0 and dist < 2 + 0 and (k - 1 - x) < window:
                t = insts[x][1]
                if t == "LABEL":
                    x -= 1; continue            # a join: predecessors unknown, keep looking along the layout
                op = t.split()[0]
                w = set()
                if op in ("v_readlane_b32", "v_readfirstlane_b32"):
                    w = sregs(t.split(None, 1)[1].split(",")[0])
                elif op.startswith("v_cmp") and op.endswith("_e64"):
                    w = sregs(t.split(None, 1)[1].split(",")[0])
                elif op.startswith("v_") and "vcc" in t.split(None, 1)[1].split(",")[0:2][-1] and op.startswith(("v_add_co", "v_sub_co", "v_mad_u64")):
                    w = {-1}
                if w & reads:
                    bad += 1
                    print("line %d: asm block reads %s written by VALU %d wait state(s) earlier: %s" % (insts[k][0] + 1, sorted(w & reads), dist, t))
                    break
                m2 = re.match(r"s_nop (\d+)", t)
                dist += (int(m2.group(1)) + 1) if m2 else 1
                x -= 1
        k = j
    print("%s: %d asm blocks with scalar inputs checked, %d inside the hazard window" % (path, checked, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
