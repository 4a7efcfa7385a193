"""In-kernel phase timing of k_extend_mp (profiling build of the library)."""
import ctypes as C, os, sys, numpy as np, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smcsmc_amd import pf, build as _build
pf.LIB_PATH = _build.build_stamps_lib()          # the -DPF_STAMPS build of the library (smcsmc_amd/build.py)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=3000); ap.add_argument("--epochs", type=int, default=32)
ap.add_argument("--np", type=int, default=20000); ap.add_argument("--debug", type=int, default=0)
a = ap.parse_args()
args = argparse.Namespace(nsam=8, length=3e6, epochs=a.epochs, pops=2)
model, segs = bench.build_workload(args, seed=1)
f = pf.ParticleFilter(model, a.np, seed=1, max_trace_events=0, local_recomb=True, debug=a.debug)
f.load_segments(segs)
L = f.L
L.pf_debug_stamps.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]; L.pf_debug_stamps.restype = C.c_int
rows = min(a.rows, len(segs["start"]))
f.init_prior(0.0); f.run(0, 200); f.sync()
assert L.pf_debug_stamps(f.h, rows, None) == 0
f.init_prior(0.0); f.run(0, rows); f.sync()
nc = (a.np + 63) // 64
out = np.zeros((rows, nc, 32), np.uint64)
assert L.pf_debug_stamps(f.h, rows, out.ctypes.data) == 0
us = out[50:].astype(np.int64) * 0.01
names = ["load state", "pop at cut", "walk prologue", "walk loop", "walk flush", "slots+choice", "edit lists+insert", "tree length",
         "weight+sample point+record head", "tracked len+next base", "site likelihood", "stores", "outer walk loops", "stretches", "update trips", "TOTAL"]
tot = us[:, :, 15]
crit = tot.argmax(axis=1)
print("rows", rows, "mean wave total %.1f us, max over waves (mean over rows) %.1f us" % (tot.mean(), tot.max(axis=1).mean()))
for k in range(16):
    col = us[:, :, k]
    c = col[np.arange(len(crit)), crit]
    if k in (12, 13, 14):
        print("   %-34s mean %7.2f   critical wave %7.2f   (count)" % (names[k], col.mean() * 100, c.mean() * 100))
    else:
        print("   %-34s mean %7.2f   critical wave %7.2f us" % (names[k], col.mean(), c.mean()))

cyc = out[50:].astype(np.int64)
cn = ["outer: flush+predraw", "iter top (search, need)", "fire: epoch loop", "fire: kind+record", "quiet: record+advance"]
for k in range(16, 21):
    col = cyc[:, :, k]
    c = col[np.arange(len(crit)), crit]
    print("   %-34s mean %9.0f   critical wave %9.0f  cycles (s_memtime)" % (cn[k - 16], col.mean(), c.mean()))

pn = ["  load: tables in LDS, barrier", "  load: decision on the previous row", "  load: parent search (resampling rows)", "  load: tree, event lists", "  load: rest (weights, completion, row data)"]
for k in range(21, 26):
    col = us[:, :, k]
    c = col[np.arange(len(crit)), crit]
    print("   %-42s mean %7.2f   critical wave %7.2f us" % (pn[k - 21], col.mean(), c.mean()))
