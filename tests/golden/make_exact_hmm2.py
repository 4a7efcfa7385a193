#!/usr/bin/env python3
"""Generates tests/golden/exact_hmm2.json: the exact E-step of the two-sample SMC' model (tests/exact_hmm2.py: forward-backward
on a discretised coalescence time, no reference code, no oracle) on the reference's committed two-sample data sets, at the
parameters the reference's regression classes start from (tests/golden/reference_bands.json).

    python tests/golden/make_exact_hmm2.py [K]        (CPU only; about two minutes at the default K = 600)

The fixture holds, per class: the exact log-likelihood, the expected coalescence count / opportunity and recombination count /
opportunity per epoch, the estimates they imply (with the pseudo-counts of count.cpp:161-227, as the .out file has them), and
the same at half the grid resolution (what `refine` reports: the discretisation error).  tests/test_exact_hmm_cpu.py and
tests/test_gpu_exact_hmm.py hold the oracle and the device path to these numbers.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import exact_hmm2  # noqa: E402
from smcsmc_amd import segments as segmod  # noqa: E402

CLASSES = ["TestConstPopSize", "TestConstPopSize_FourEpochs", "TestConstPopSize_FourEpochs_FalseStart"]


def case_inputs(name):
    """model and packed rows of a reference class, as the binary's host side reads them (-dumpmodel: CPU only)"""
    cases = json.load(open(os.path.join(ROOT, "tests/golden/reference_bands.json")))["cases"]
    c = [x for x in cases if x["name"] == name][0]
    seg = os.path.join(ROOT, "tests/golden/seg", c["data"])
    argv = [seg if a == "@SEG@" else a for a in c["binary_argv"]]
    out = subprocess.run([os.path.join(ROOT, "bin/smcsmc")] + argv + ["-dumpmodel"], capture_output=True, text=True)
    m = json.loads(out.stdout.splitlines()[-1])
    assert m["nsam"] == 2 and len(np.array(m["pop_sizes"]).shape) == 2
    rows = segmod.Segments(seg, 2, m["loci_length"], max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    E = len(m["change_times"])
    packed = rows.pack(np.full(E, 1e99))          # record limit: every epoch at every row (true for these files at the calibrated lags)
    return c, m, packed


def exact(m, packed, K):
    ne = np.array(m["pop_sizes"], float)[:, 0]
    h = exact_hmm2.ExactHMM2(m["change_times"], ne, m["mutation_rate"], m["recombination_rate"], K=K)
    res = h.run(packed, seq_len=m["loci_length"])
    ne_hat, rho_hat = exact_hmm2.estimates(res, ne, m["recombination_rate"])
    return dict(K=h.g.K, logl=res["logl"], coal_count=res["coal_count"].tolist(), coal_opp=res["coal_opp"].tolist(),
                rec_count=res["rec_count"].tolist(), rec_opp=res["rec_opp"].tolist(), ne=ne_hat.tolist(), rho=float(rho_hat))


REPLICATES = dict(n=40, first_seed=1000, L=1.0e7, mu=2.5e-8, rho=1.0e-8, change_times=[0.0, 400.0, 10000.0, 20000.0, 40000.0, 60000.0], ne=1.0e4, K=250)


def replicate_rows(rep):
    """data set `rep` of the replicate test: two samples, 10 Mb, from the numpy simulator (smcsmc_amd/simulate.py), packed as a .seg file is"""
    from smcsmc_amd import simulate
    R = REPLICATES
    ct, ne = np.array(R["change_times"]), np.full(len(R["change_times"]), R["ne"])
    seg = simulate.simulate_seg(2, R["L"], R["mu"], R["rho"], ct, ne, seed=R["first_seed"] + rep)
    S = segmod.Segments.from_sites(seg["start"], seg["length"], seg["alleles"], 2, R["L"], max_segment_length=int(2.0 / (R["rho"] * 4 * R["ne"])))
    return S


def replicates():
    """the exact E-step summed over the forty simulated data sets of tests/test_gpu_exact_hmm.py::test_forty_replicates...: what the
    filter must reproduce there.  (It is NOT the truth: the reference's emission -- no mutation over the whole row, then the site --
    books the site's base twice, which costs the deep epochs 0.7-0.9 % on infinite-sites data; the exact E-step shows that offset
    as the filter does.)"""
    R = REPLICATES
    ct, ne = np.array(R["change_times"]), np.full(len(R["change_times"]), R["ne"])
    h = exact_hmm2.ExactHMM2(ct, ne, R["mu"], R["rho"], K=R["K"])
    tot = {k: np.zeros(len(ct)) for k in ("coal_count", "coal_opp", "rec_count", "rec_opp")}
    logl = 0.0
    for rep in range(R["n"]):
        res = h.run(replicate_rows(rep).pack(np.full(len(ct), 1e99)), seq_len=R["L"])
        for k in tot:
            tot[k] += res[k]
        logl += res["logl"]
        print("replicate", rep, " ".join("%.0f" % v for v in tot["coal_opp"] / (2 * tot["coal_count"])), flush=True)
    return dict(R, pooled={k: v.tolist() for k, v in tot.items()}, pooled_ne=(tot["coal_opp"] / (2 * tot["coal_count"])).tolist(),
                pooled_rho=float(tot["rec_count"].sum() / tot["rec_opp"].sum()), logl_sum=logl)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    out = {"generator": "tests/golden/make_exact_hmm2.py", "method": "tests/exact_hmm2.py", "classes": {}}
    if "--replicates-only" in sys.argv:
        out = json.load(open(os.path.join(ROOT, "tests/golden/exact_hmm2.json")))
        out["replicates"] = replicates()
        json.dump(out, open(os.path.join(ROOT, "tests/golden/exact_hmm2.json"), "w"), indent=1)
        return
    for name in CLASSES:
        c, m, packed = case_inputs(name)
        fine, coarse = exact(m, packed, K), exact(m, packed, K // 2)
        out["classes"][name] = dict(data=c["data"], change_times=m["change_times"], start_ne=np.array(m["pop_sizes"])[:, 0].tolist(),
                                    mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"],
                                    rows=int(len(packed["start"])), exact=fine, half_resolution=coarse,
                                    bands=[dict(type=t["type"], epoch=t.get("epoch"), min=t["min"], max=t["max"]) for t in c["targets"]])
        print(name, "logl %.3f" % fine["logl"], " ".join("%.1f" % v for v in fine["ne"]), "rho %.5e" % fine["rho"],
              "| K/2:", " ".join("%.1f" % v for v in coarse["ne"]), "%.5e" % coarse["rho"], flush=True)
    out["replicates"] = replicates()
    with open(os.path.join(ROOT, "tests/golden/exact_hmm2.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
