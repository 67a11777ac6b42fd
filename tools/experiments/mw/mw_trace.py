"""Timeline of chain 0 of the multi-wave per-draw kernel (library built with -DMW_TRACE by tools/experiments/mw_trace.sh and
selected through DHMC_LIB_PATH): clock deltas between the instrumented events of the vector waves and the control wave."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.abi.lib()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = pkg.DeviceContext(1000, C, seed=1)
ctx.init(); ctx.set_stepsize(0.3)
ctx.run(5, fields=[])
buf = np.zeros((8, 4096), np.uint64); cnt = np.zeros(8, np.uint32)
lib.dhmc_debug_mw_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.dhmc_debug_mw_trace(None, None, 1)
ctx.run(6, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
lib.dhmc_debug_mw_trace(buf.ctypes.data, cnt.ctypes.data, 0)
print("chains", C, "kernel_ms", ms, "leapfrogs", lf, "counts", cnt.tolist())
ev = {}
for w in range(5):
    n = min(int(cnt[w]), 4096)
    e = (buf[w, :n] >> np.uint64(56)).astype(int); t = (buf[w, :n] & np.uint64((1 << 56) - 1)).astype(np.int64)
    ev[w] = (e, t)
t0 = min(ev[w][1][0] for w in range(5) if len(ev[w][1]))
span = max(ev[w][1][-1] for w in range(5) if len(ev[w][1])) - t0
print("clock span of the traced run", span, "-> clocks per ms", span / ms, " clocks per leapfrog", span / (lf / C))
# per-wave: mean delta from each event kind to the next event
for w in range(5):
    e, t = ev[w]
    if len(e) < 2: continue
    d = np.diff(t)
    print("wave", w, "(control)" if w == 4 else "")
    for a in sorted(set(e[:-1].tolist())):
        for b in sorted(set(e[1:].tolist())):
            m = (e[:-1] == a) & (e[1:] == b)
            if m.sum():
                print(f"   {a:2d} -> {b:2d}: n={int(m.sum()):4d} mean={d[m].mean():8.0f} min={d[m].min():6d} max={d[m].max():6d}")
# first 60 events of wave 0 and control, merged by time
allev = sorted([(int(t), w, int(e)) for w in (0, 4) for e, t in zip(*ev[w])])[:90]
print("merged timeline (clock - t0, wave, event):")
print("  ".join(f"{t - t0}:{'C' if w == 4 else 'V'}{e}" for t, w, e in allev))
