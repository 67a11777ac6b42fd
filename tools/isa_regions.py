"""Static view of a kernel's ISA: per source region (the PH(i) marks of nuts_kernels.hpp compiled with -DDHMC_PHASE_MARK)
the number of instructions by class, and where scratch (spill) traffic sits.  No GPU needed:

    python tools/isa_regions.py NPL [extra hipcc flags]        (e.g. 16; FAM=DiagNormalT in the environment)

compiles tools/experiments/kernel_only.hip (one instantiation of nuts_run_kernel) to assembly and prints the instruction mix of each region in LAYOUT order (a region = the code between two marks; the
compiler lays blocks out roughly in source order, so this is indicative, not a dynamic count).
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "outside", 1: "momentum+setup", 2: "edge switch", 3: "leaf", 4: "leaf scalars", 5: "merge vector",
         6: "merge scalar", 7: "suspend", 8: "end of transition", 9: "write back"}


def classify(op):
    if op.startswith("v_accvgpr"): return "agpr"
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): return "lane"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.endswith("_dpp") or "dpp" in op: return "dpp"
    if op.startswith("v_mov"): return "vmov"
    if re.match(r"v_(add|mul|fma|fmac|max|min|ldexp|div|rcp|rsq|sqrt|frexp|floor|fract|trunc|rndne)\w*_f64", op) or op.startswith("v_div_"): return "f64"
    if op.startswith("v_cmp"): return "vcmp"
    if op.startswith("v_"): return "vint"
    if op.startswith("scratch"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global") or op.startswith("buffer") or op.startswith("flat"): return "vmem"
    if op == "s_waitcnt" or op == "s_nop": return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    npl = sys.argv[1]
    extra = sys.argv[2:]
    pat = ""
    fam = os.environ.get("FAM", "StdNormalT")
    out = os.environ.get("ISA_OUT", "/tmp/isa_regions_%s_%s.s" % (fam, npl))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result",
           "-Wno-unused-command-line-argument", "-DKO_T=" + fam, "-DKO_NPL=" + npl, "--cuda-device-only", "-S", "-o", out] + extra + ["kernel_only.hip"]
    if "-DNO_MARK" not in extra:
        cmd.insert(1, "-DDHMC_PHASE_MARK")
    subprocess.check_call(cmd, cwd=os.path.join(ROOT, "tools", "experiments"))
    cur = None
    funcs = {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur:
            funcs[cur].append(line.rstrip("\n"))
    cols = ["f64", "vint", "vcmp", "vmov", "dpp", "lane", "cndmask", "agpr", "salu", "branch", "lds", "vmem", "smem", "scratch", "wait"]
    for name, lines in funcs.items():
        if "nuts_run_kernel" not in name or pat not in name:
            continue
        region = 0
        counts = collections.OrderedDict()
        order = []
        for l in lines:
            t = l.strip()
            m = re.match(r"; DHMC_PH (\d+)", t)
            if m:
                region = int(m.group(1))
                continue
            if not l.startswith("\t") or t.startswith(".") or t.startswith(";") or not t:
                continue
            op = t.split()[0]
            key = region
            if key not in counts:
                counts[key] = collections.Counter()
            counts[key][classify(op)] += 1
        print(name)
        print("%-20s %6s " % ("region", "all") + " ".join("%7s" % c for c in cols))
        tot = collections.Counter()
        for key, c in counts.items():
            tot.update(c)
            print("%-20s %6d " % (NAMES.get(key, str(key)), sum(c.values())) + " ".join("%7d" % c[k] for k in cols))
        print("%-20s %6d " % ("total", sum(tot.values())) + " ".join("%7d" % tot[k] for k in cols))
        for l in lines:
            if re.search(r"\.(sgpr|vgpr)_(count|spill_count)|scratch_size|; (NumVgprs|NumAgprs|ScratchSize|Occupancy)", l):
                print("   ", l.strip())


if __name__ == "__main__":
    main()
