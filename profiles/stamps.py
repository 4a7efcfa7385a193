"""In-kernel phase timing of the extend workgroups (profiling build of the library)."""
import ctypes as C, os, sys, numpy as np, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smcsmc_amd import pf, build as _build
pf.LIB_PATH = _build.build_stamps_lib()          # the -DPF_STAMPS build of the library (smcsmc_amd/build.py)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=6000); ap.add_argument("--debug", type=int, default=0)
a = ap.parse_args()
args = argparse.Namespace(nsam=4, length=1e7, epochs=32, pops=1)
model, segs = bench.build_workload(args, seed=1)
f = pf.ParticleFilter(model, 10000, seed=1, max_trace_events=0, local_recomb=True, debug=a.debug)
f.load_segments(segs)
L = f.L
L.pf_debug_stamps.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]; L.pf_debug_stamps.restype = C.c_int
f.init_prior(0.0); f.run(0, 200); f.sync()      # warm up
rows = a.rows
assert L.pf_debug_stamps(f.h, rows, None) == 0
f.init_prior(0.0); f.run(0, rows); f.sync()
nc = (10000 + 63) // 64
out = np.zeros((rows, nc, 32), np.uint64)
assert L.pf_debug_stamps(f.h, rows, out.ctypes.data) == 0
st = out[5:, :, :9].astype(np.int64) * 10.0 / 1000.0      # microseconds
flag = f.trace()["resampled"][:rows]
t0 = st[:, :, 0].min(axis=1, keepdims=True)
rel = st - t0[:, :, None]
names = ["start", "loads issued", "decide_row done", "prologue done", "state ready", "update loop done", "site lik done", "stores issued", "scans done"]
end = rel[:, :, 8].max(axis=1)
print("rows", rows, "kernel span (first wave start -> last wave scans done): mean %.2f us" % end.mean())
for sel, nm in ((np.ones(len(end), bool), "all rows"), (flag[4:rows-1] == 1, "rows after a resampling row"), (flag[4:rows-1] == 0, "rows after a plain row")):
    sel = sel[:len(end)]
    print(nm, int(sel.sum()))
    for k in range(9):
        col = rel[sel][:, :, k]
        print("   %-18s mean over waves %6.2f   max over waves (mean over rows) %6.2f" % (names[k], col.mean(), col.max(axis=1).mean()))
    # the critical wave: phase durations of the wave that finishes last
    cr = rel[sel]
    idx = cr[:, :, 8].argmax(axis=1)
    crit = cr[np.arange(len(idx)), idx, :]
    print("   critical wave phase durations:", " ".join("%.2f" % v for v in np.diff(crit, axis=1).mean(axis=0)), " start offset %.2f" % crit[:, 0].mean())

# the parent search of the rows that follow a resampling row, step by step (stamps 16-20 between "decide_row done" and "prologue done")
sub = out[5:, :, 16:21].astype(np.int64) * 10.0 / 1000.0
selr = (flag[4:rows-1] == 1)[:len(end)]
if selr.any() and sub[selr].max() > 0:
    base = st[selr][:, :, 2]
    seq = [base] + [sub[selr][:, :, k] for k in range(5)] + [st[selr][:, :, 3]]
    nm = ["own offset (pipe_lo_from)", "search over wavefronts", "barrier, range, survivors", "staging decided, barrier", "search inside the wavefront", "first copy?, offspring table, barrier"]
    print("parent search, rows after a resampling row (mean over wavefronts of the step's duration):")
    for k in range(6):
        d = seq[k + 1] - seq[k]
        print("   %-40s %6.2f us" % (nm[k], d.mean()))

# What a wavefront does between "state ready" and "scans done" needs nothing from other wavefronts.  Were two rows run back to
# back per wavefront (the second one speculating that the first does not resample), a pair would take as long as the
# wavefront with the largest SUM, not the sum of the two largest:
d = rel[:, :, 8] - rel[:, :, 4]
m = (len(d) // 2) * 2
pair = (d[0:m:2] + d[1:m:2]).max(axis=1)
single = d[0:m:2].max(axis=1) + d[1:m:2].max(axis=1)
pre = rel[:, :, 4].max(axis=1)
print("own work of a row (state ready -> scans done), slowest wavefront: %.2f us; two rows back to back, slowest wavefront: %.2f us per pair"
      " (two single rows: %.2f us); before it (launch start -> state ready, slowest wavefront): %.2f us" % (d.max(axis=1).mean(), pair.mean(), single.mean(), pre.mean()))
for k in (3, 4):
    mk = (len(d) // k) * k
    grp = sum(d[i:mk:k] for i in range(k)).max(axis=1)
    print("   %d rows back to back: %.2f us per group" % (k, grp.mean()))

acc = out[5:, :, 9:15].astype(np.int64)
trips = acc[:, :, 5].astype(float)
print("update trips per wave and row: mean %.2f, max over waves (mean over rows) %.2f" % (trips.mean(), trips.max(axis=1).mean()))
tot = acc[:, :, :5].sum(axis=(0, 1)) * 10.0 / 1000.0
names2 = ["no-mutation weight (all iterations)", "record head stores", "genealogy update", "record tail + tracked length", "next recombination position"]
for nm, v in zip(names2, tot):
    print("   %-40s %.3f us per trip" % (nm, v / max(1.0, trips.sum())))
