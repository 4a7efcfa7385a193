"""One-off: the C5 shape at Np = 20 000 over several hundred rows against the oracle."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, oracle_lib as oracle
import test_gpu_headline as th
L = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0e5
model = th._bench_model(8, 32, L, pops=2)
segs = cases.make_segments(dict(model, pop_sizes=model["pop_sizes"][:, 0]), seed=2, max_seg_len=5000)
print("rows", len(segs["start"]), flush=True)
t0 = time.time()
to, co, g = th._compare_sweep(oracle, model, segs, 20000, seed=2, structured=True)
print("identical: T, ess, logl, resampling flags and indices, particle states, migration events; counts within 1e-9; %d rows, %d resampling rows, %.0f migration events counted, %.0f s"
      % (len(to["T"]), int(to["resampled"].sum()), float(co["mig_count"].sum()), time.time() - t0), flush=True)
