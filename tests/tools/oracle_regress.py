"""CPU bisect harness: the reference's regression configurations through the CPU oracle .

    python tests/tools/oracle_regress.py <case> <seeds> ["dict(np_override=..., lag_fraction=..., nobias=True, record_all=True)"]

The oracle is bit-identical to the device path, so offsets against the reference's bands can be bisected without a GPU
(DESIGN.md section 6)."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from smcsmc_amd import segments as segmod

def setup(case_name, np_override=None, lag_fraction=None, nobias=False, record_all=False, delay=0.5, strengths=None, delay_type=0):
    cases = json.load(open(os.path.join(ROOT, "tests/golden/reference_bands.json")))["cases"]
    c = [x for x in cases if x["name"] == case_name][0]
    argv = [a for a in c["binary_argv"]]
    seg = os.path.join(ROOT, "tests/golden/seg", c["data"])
    argv[argv.index("@SEG@")] = seg
    m = json.loads(subprocess.run([os.path.join(ROOT, "bin/smcsmc")] + argv + ["-dumpmodel"], capture_output=True, text=True).stdout.splitlines()[-1])
    E = len(m["change_times"])
    def opt(name, k=1):
        i = argv.index(name); return argv[i + 1:i + 1 + k]
    if "-bias_heights" not in argv: nobias = True
    bh = [float(opt("-bias_heights")[0])] if not nobias else []
    bs = [float(x) for x in opt("-bias_strengths", 2)] if not nobias else []
    lf = float(opt("-calibrate_lag")[0]) if lag_fraction is None else lag_fraction
    model = dict(change_times=np.array(m["change_times"], float), pop_sizes=np.array(m["pop_sizes"], float)[:, 0],
                 nsam=m["nsam"], loci_length=float(m["loci_length"]), mutation_rate=m["mutation_rate"],
                 recombination_rate=m["recombination_rate"])
    ct = model["change_times"]
    model["lags"] = np.array([4.0 / (model["recombination_rate"] * (ct[e + 1] if e + 1 < E else ct[-1])) for e in range(E)])
    med, trees = oracle_lib.median_survival(model, seed=1, min_events=200, max_trees=1000000)
    model["lags"] = med * lf
    if not nobias:
        model.update(bias_heights=bh, bias_strengths=strengths or bs, application_delays=med * delay, delay_type=delay_type)
    S = segmod.Segments(seg, m["nsam"], m["loci_length"], max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    if record_all: segs['max_record_epoch'][:] = E - 1
    return c, model, segs, (np_override or c["np"])

def run_one(args):
    case_name, seed, kw = args
    oracle_lib.build()
    c, model, segs, Np = setup(case_name, **kw)
    o = oracle_lib.Oracle(model, Np, seed=seed, max_trace_events=0)
    o.init_prior(segs["start"][0]); si = o.pack_segments(model, segs); o.run(si)
    cn = o.counts()
    E = len(model["change_times"])
    N0 = model["pop_sizes"]
    # add the reference's pseudo counts (count.cpp:161-227): coal count 1/(2Ne), opp 1; recomb count rho, opp 1
    ne = (cn["coal_opp"] + 1.0) / (2 * (cn["coal_count"] + 1.0 / (2 * N0)))
    rec = (cn["rec_count"].sum() + model["recombination_rate"]) / (cn["rec_opp"].sum() + 1.0)
    tr = o.trace()
    return list(ne) + [rec], float(o.logl()), float(np.mean(tr["resampled"])), float(np.mean(tr["ess"]))

if __name__ == "__main__":
    import multiprocessing as mp
    case = sys.argv[1]; nseeds = int(sys.argv[2]); kw = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
    t = time.time()
    with mp.Pool(min(8, nseeds)) as p:
        res = p.map(run_one, [(case, s, kw) for s in range(1, nseeds + 1)])
    a = np.array([r[0] for r in res])
    print(case, kw, "%.0fs" % (time.time() - t))
    print(" mean", " ".join("%.0f" % v for v in a.mean(0)[:-1]), "rec %.4e" % a.mean(0)[-1], "logl %.1f" % np.mean([r[1] for r in res]))
    print(" sd  ", " ".join("%.0f" % v for v in a.std(0)[:-1]), "rec %.2e" % a.std(0)[-1])
    print(" resampling rows %.3f, mean ESS %.1f" % (np.mean([r[2] for r in res]), np.mean([r[3] for r in res])))
