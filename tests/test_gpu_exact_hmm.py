"""The device path (bin/smcsmc: C++ host, C-ABI, HIP kernels) against the exact E-step of the two-sample model.

tests/golden/exact_hmm2.json holds what forward-backward on a fine grid of coalescence times gives for the reference's
constpopsize.seg / constpopsize_4epochs.seg at the reference's command-line parameters (tests/exact_hmm2.py; no reference code,
no oracle).  The filter is a Monte Carlo approximation of exactly that E-step, so its estimates must approach those numbers as the
number of particles grows -- with and without focused sampling (-bias_heights 400 -bias_strengths 3 1, what TestConstPopSize
runs), whose importance weights must leave the target unchanged.  The lag is four survival distances (the exact numbers are
fully smoothed).  profiles/round4/exact_hmm2.md has the table over Np = 1 000 / 4 000 / 16 000.
"""
import json
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import hmm_convergence as hc   # noqa: E402
import reference_bands as rb   # noqa: E402

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(ROOT, "tests/golden/exact_hmm2.json")))["classes"]
SEEDS = [1, 2, 3, 4, 5, 6]


@pytest.fixture(scope="module")
def runs(hiplib):
    cases = {c["name"]: c for c in rb.load_cases(variants=False)}
    tmp = tempfile.mkdtemp(prefix="exacthmm_")
    out = {}
    for name, focus, Np in [("TestConstPopSize", False, 1000), ("TestConstPopSize", False, 16000),
                            ("TestConstPopSize", True, 1000), ("TestConstPopSize", True, 16000),
                            ("TestConstPopSize_FourEpochs", False, 16000)]:
        out[(name, focus, Np)] = hc.run(cases[name], focus, Np, 4.0, SEEDS, tmp)
    return out


def _rel(runs, name, focus, Np):
    ex = GOLD[name]["exact"]
    exact = np.array(ex["ne"] + [ex["rho"], ex["logl"]])
    return runs[(name, focus, Np)].mean(0) / exact - 1.0


@pytest.mark.parametrize("name,focus", [("TestConstPopSize", False), ("TestConstPopSize", True), ("TestConstPopSize_FourEpochs", False)])
def test_estimates_at_16000_particles_are_the_exact_ones(runs, name, focus):
    """Ne of every epoch with data and rho within 0.5 % of the exact expected-count ratios, the log-likelihood within 1e-4
    relative (mean of six seeds; measured: 0.2 % and 1e-5)"""
    rel = _rel(runs, name, focus, 16000)
    E = len(GOLD[name]["exact"]["ne"])
    assert np.abs(rel[1:E]).max() < 5e-3, rel
    assert abs(rel[E]) < 5e-3 and abs(rel[E + 1]) < 1e-4, rel


@pytest.mark.parametrize("focus", [False, True])
def test_the_offset_shrinks_with_the_number_of_particles(runs, focus):
    """what is left at Np = 1 000 (up to -0.75 % in the oldest epoch) is a finite-Np effect: it is smaller at Np = 16 000"""
    name = "TestConstPopSize"
    E = len(GOLD[name]["exact"]["ne"])
    r1, r16 = _rel(runs, name, focus, 1000), _rel(runs, name, focus, 16000)
    rms = lambda r: float(np.sqrt(np.mean(r[1:E + 1] ** 2)))
    assert rms(r16) < rms(r1) and rms(r1) < 1.2e-2, (r1, r16)
    assert abs(r16[E + 1]) < abs(r1[E + 1]) + 2e-6


def test_the_reference_bands_of_these_classes_are_not_the_exact_values():
    """what the exact E-step says about the targets of test_const_pop_size.py:42-49 that this build misses: the bands of
    epochs 1 and 2 and of the recombination rate do not contain the exact value (they were taken from runs of a reference
    binary at Np = 1 000, not from the model), the build's values do agree with it"""
    ex = GOLD["TestConstPopSize"]
    outside = [b for b in ex["bands"] if not (b["min"] <= (ex["exact"]["rho"] if b["type"] == "Recomb" else ex["exact"]["ne"][b["epoch"]]) <= b["max"])]
    assert {(b["type"], b["epoch"]) for b in outside} >= {("Coal", 1), ("Coal", 2), ("Recomb", None)}


def test_forty_replicates_equal_the_exact_e_step_and_sit_near_the_truth(hiplib):
    """Forty independent 10 Mb data sets of two samples from the numpy simulator (smcsmc_amd/simulate.py, independent of the device
    code), each filtered once at the true parameters (Np = 16 000, lag of four survival distances), the sufficient statistics summed
    over the data sets.  Two things are held:
    (1) the pooled ratios equal those of the EXACT E-step summed over the same forty data sets (tests/golden/exact_hmm2.json,
        "replicates") to 0.3 % for every epoch with data and 0.1 % for rho -- forty times the evidence of the single-file tests;
    (2) they sit within 1.2 % of the truth.  Not closer, and the exact E-step says why: the reference's emission (no mutation over
        the whole row, then the site likelihood: the site's base is booked twice, and the two-state site model allows back
        mutation) is an approximation of order mu x T, which on infinite-sites data costs the epochs beyond 40 000 generations
        0.7 - 0.9 % -- in the exact E-step and in the filter alike (with the standard error of the forty data sets at 0.2 %, a
        test of the filter against the truth alone fails at five standard errors, at Np = 4 000 and at Np = 16 000)."""
    from smcsmc_amd import ParticleFilter, pf
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_exact_hmm2 as mk
    R = json.load(open(os.path.join(ROOT, "tests/golden/exact_hmm2.json")))["replicates"]
    assert (R["n"], R["first_seed"]) == (mk.REPLICATES["n"], mk.REPLICATES["first_seed"])
    ct = np.array(R["change_times"]); E = len(ct)
    ne = np.full(E, R["ne"])
    model = dict(change_times=ct, pop_sizes=ne, nsam=2, loci_length=R["L"], mutation_rate=R["mu"], recombination_rate=R["rho"], lags=np.ones(E))
    med, _ = pf.median_survival(model, seed=1, min_events=200, max_trees=1000000)
    model["lags"] = med * 4.0
    tot = {k: np.zeros(E) for k in ("coal_count", "coal_opp", "rec_count", "rec_opp")}
    for rep in range(R["n"]):
        segs = mk.replicate_rows(rep).pack(model["lags"])
        g = ParticleFilter(model, 16000, seed=rep + 1, max_trace_events=0)
        g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
        c = g.counts()
        for k in tot:
            tot[k] += c[k]
        g.close()
    ne_hat = tot["coal_opp"] / (2 * tot["coal_count"])
    rho_hat = tot["rec_count"].sum() / tot["rec_opp"].sum()
    exact_ne, exact_rho = np.array(R["pooled_ne"]), R["pooled_rho"]
    np.testing.assert_allclose(ne_hat[1:], exact_ne[1:], rtol=3e-3)
    assert abs(ne_hat[0] / exact_ne[0] - 1) < 0.05                       # (forty events in all: the filter's own noise)
    assert abs(rho_hat / exact_rho - 1) < 1e-3
    assert np.abs(ne_hat[1:] / ne[1:] - 1).max() < 1.2e-2 and abs(rho_hat / R["rho"] - 1) < 5e-3
    assert np.abs(exact_ne[1:] / ne[1:] - 1).max() < 1.2e-2             # the exact E-step carries the same offset from the truth
