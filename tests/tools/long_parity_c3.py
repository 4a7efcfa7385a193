"""One-off: the C3 shape at Np = 10 000 over ~3 000 rows against the oracle (ten times the test's length)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, oracle_lib as oracle
import test_gpu_headline as th
L = float(sys.argv[1]) if len(sys.argv) > 1 else 1.6e6
model = th._bench_model(4, 32, L)
segs = cases.make_segments(model, seed=1, max_seg_len=5000)
print("rows", len(segs["start"]), flush=True)
t0 = time.time()
to, co, g = th._compare_sweep(oracle, model, segs, 10000, seed=1)
print("identical: T, ess, logl, resampling flags and indices, particle states; counts within 1e-9; %d rows, %d resampling rows, logl %.6f, %.0f s"
      % (len(to["T"]), int(to["resampled"].sum()), float(to["logl"][-1]) if len(to["logl"]) else 0.0, time.time() - t0), flush=True)
