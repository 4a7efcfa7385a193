"""CPU tests of the oracle itself (no GPU): math accuracy, RNG known answers, canonical reductions,
coalescent prior expectations and a distributional acceptance run on reference data.

The reference's own tests hold no golden vectors for weights / indices / log-likelihood
("parity unpinned"), so the oracle is pinned by what *can* be pinned:
published known answers (Philox), analytic expectations of the model it simulates, the
reference's committed real-scrm data with the known simulation truth, and the acceptance bands of
the reference's own no-data regression classes (tests/golden/reference_bands.json; the other
classes need Np = 1000 over 10 Mb and run on the GPU, tests/test_gpu_reference_bands.py)."""
import numpy as np
import pytest

import cases


def test_exp_log_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-700, 700, 5000), rng.uniform(-1, 1, 5000), [-745.2, 709.9, 0.0]])
    e = np.array([L.smco_exp(v) for v in x])
    ref = np.exp(x)
    ok = np.isfinite(ref) & (ref > 1e-300)
    assert np.max(np.abs(e[ok] / ref[ok] - 1)) < 4e-16
    assert L.smco_exp(710.0) == np.inf and L.smco_exp(-746.0) == 0.0
    y = np.concatenate([rng.uniform(0, 1, 5000), 10.0 ** rng.uniform(-300, 300, 5000), [1.0, 5e-324]])
    lg = np.array([L.smco_log(v) for v in y])
    refl = np.log(y)
    nz = np.abs(refl) > 1e-6
    assert np.max(np.abs(lg[nz] / refl[nz] - 1)) < 4e-16
    assert L.smco_log(1.0) == 0.0


def test_fastexp_matches_reference_formula(oracle):
    """particle.cpp:30-40: rational approximation for x^2 < 0.516167859, exp otherwise; rel. error < 1e-6."""
    L = oracle.lib()
    for x in np.linspace(-3, 1, 401):
        f = L.smco_fastexp(float(x))
        if x * x < 0.516167859:
            assert f == 1 + 2 * x / (2 - x + x * x / (6 + x * x * 0.1))
        assert abs(f / np.exp(x) - 1) < 1.1e-6


def test_philox_known_answer(oracle):
    """Philox4x32-10, counter = key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8 (Random123 kat_vectors)."""
    L = oracle.lib()
    u = L.smco_uniform(0, 0, 0, 0)
    bits = ((0x6627e8d5 << 32) | 0xe169c58d) >> 11
    assert u == (bits + 0.5) * 2.0 ** -53
    # counter = ffffffff x4, key = ffffffff x2 -> 408f276d 41c83b0e a20bc7c6 6d5451fd
    u = L.smco_uniform(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF)
    bits = ((0x408f276d << 32) | 0x41c83b0e) >> 11
    assert u == (bits + 0.5) * 2.0 ** -53


def test_canonical_sum_and_scan_definitions(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for n in (1, 5, 64, 65, 4096, 4097, 10000):
        x = rng.exponential(1.0, n)
        s = L.smco_canon_sum(x.ctypes.data, n)
        assert abs(s - x.sum()) <= 1e-12 * x.sum()
        sc = np.zeros(n); L.smco_canon_scan(x.ctypes.data, sc.ctypes.data, n)
        np.testing.assert_allclose(sc, np.cumsum(x), rtol=1e-13)
    # pairwise tree of the first chunk, written out by hand for n = 4: (x0+x1)+(x2+x3)
    x = np.array([0.1, 0.2, 0.3, 0.4])
    assert L.smco_canon_sum(x.ctypes.data, 4) == (x[0] + x[1]) + (x[2] + x[3])
    sc = np.zeros(4); L.smco_canon_scan(x.ctypes.data, sc.ctypes.data, 4)
    assert sc[3] == (x[0] + x[1]) + (x[2] + x[3]) and sc[2] == (x[0]) + (x[1] + x[2]) and sc[1] == x[0] + x[1]


def test_systematic_resampling_matches_serial_reference_walk(oracle):
    """particleContainer.cpp:474-504 walked serially (with u_j = (j+U)/N) gives the same counts."""
    L = oracle.lib()
    rng = np.random.default_rng(4)
    for n in (2, 7, 64, 1000):
        w = rng.exponential(1.0, n)
        u = float(rng.uniform())
        lo = np.zeros(n + 1, np.int32)
        L.smco_systematic(w.ctypes.data, n, u, lo.ctypes.data)
        incl = np.zeros(n); L.smco_canon_scan(w.ctypes.data, incl.ctypes.data, n)
        partial = np.concatenate([[0.0], incl])
        counts = np.zeros(n, int)
        j = 0
        for k in range(n):                       # sample k has quantile (k+u)/n; pc.cpp:491 scaled by n*total
            while j + 1 < n and not (n * partial[j + 1] > (k + u) * partial[n]):
                j += 1
            counts[j] += 1
        assert (np.diff(lo) == counts).all()
        assert lo[0] == 0 and lo[n] == n


def test_prior_tree_moments(oracle):
    """E[T_k] of the Kingman coalescent: the initial trees (buildInitialTree) follow the prior."""
    N0 = 1e4
    model = cases.make_model(n=4, E=1, N0=N0)
    o = oracle.Oracle(model, 20000, seed=5)
    o.init_prior(0.0)
    H = o.particles()["heights"]
    exp = 2 * N0 * np.cumsum([1 / 6.0, 1 / 3.0, 1.0])
    np.testing.assert_allclose(H.mean(0), exp, rtol=0.03)
    ltree = 4 * H[:, 0] + 3 * (H[:, 1] - H[:, 0]) + 2 * (H[:, 2] - H[:, 1])
    assert abs(ltree.mean() / (4 * N0 * (1 + 0.5 + 1 / 3.0)) - 1) < 0.02


def test_no_data_run_reproduces_model_rates(oracle):
    """No-data mode samples the prior: counts/opportunity return the input rates,
    the likelihood is exactly 1, and nothing is resampled."""
    N0, rho = 1e4, 1e-8
    model = cases.make_model(n=4, E=1, N0=N0, rho=rho, L=4e5)
    segs = cases.nodata_segments(model)
    o = oracle.Oracle(model, 1500, seed=7)
    o.init_prior(0.0)
    o.run(o.pack_segments(model, segs))
    c = o.counts()
    assert o.logl() == 0.0 and c["resample_count"] == 0
    assert abs(c["coal_count"][0] / c["coal_opp"][0] * 2 * N0 - 1) < 0.05
    assert abs(c["rec_count"][0] / c["rec_opp"][0] / rho - 1) < 0.05
    # opportunity bookkeeping: recombination opportunity = integral of tree length, ~ L * E[tree length]
    assert abs(c["rec_opp"][0] / (4e5 * 4 * N0 * (1 + 0.5 + 1 / 3.0)) - 1) < 0.05
    assert c["delayed_opp"] == 4e5


def test_epoch_structured_prior(oracle):
    """With a bottleneck epoch the per-epoch coalescence rates follow 1/(2 N_e)."""
    model = cases.make_model(n=4, E=4, L=4e5, sizes=[1.0, 0.2, 2.0, 1.0])
    model["change_times"] = np.array([0.0, 2000.0, 8000.0, 40000.0])
    model["lags"] = np.full(4, 2e4)
    segs = cases.nodata_segments(model)
    o = oracle.Oracle(model, 1500, seed=8)
    o.init_prior(0.0)
    o.run(o.pack_segments(model, segs))
    c = o.counts()
    rate = c["coal_count"] / c["coal_opp"]
    np.testing.assert_allclose(rate * 2 * model["pop_sizes"], 1.0, rtol=0.12)


def test_inference_on_reference_data_recovers_truth(oracle):
    """Distributional acceptance on the reference's committed scrm data (test/old/newtests/testdata/
    constpopsize.seg, first 3000 rows; truth Ne = 10000, rho = 1e-8; the reference's own ranges for the
    full 10 Mb / Np 1000 run are Recomb in [9.77e-9, 9.89e-9], test_const_pop_size.py:42-49)."""
    import os
    from smcsmc_amd import segments as segmod
    path = os.path.join(os.path.dirname(__file__), "golden", "seg", "constpopsize_first3000.seg")
    N0, rho, mu = 1e4, 1e-8, 2.5e-8
    ct = np.array([0, 0.01, 0.25, 0.5, 1, 1.5]) * 4 * N0
    S = segmod.Segments(path, 2, 3e6, max_segment_length=5000)
    L = float(S.rows[-1][0] + S.rows[-1][1] - 1)
    model = dict(change_times=ct, pop_sizes=np.full(6, N0), lags=np.full(6, 2e4), nsam=2, loci_length=L,
                 mutation_rate=mu, recombination_rate=rho)
    segs = S.pack(model["lags"])
    o = oracle.Oracle(model, 300, seed=1)
    o.init_prior(0.0)
    o.run(o.pack_segments(model, segs))
    c = o.counts()
    assert c["resample_count"] > 10
    rec = c["rec_count"].sum() / c["rec_opp"].sum()
    assert 0.8e-8 < rec < 1.2e-8
    ne = c["coal_opp"].sum() / (2 * c["coal_count"].sum())
    assert 8000 < ne < 12500
    assert -1e5 < o.logl() < 0


def test_resampling_conserves_weight_and_particles(oracle):
    model = cases.make_model(n=4, E=8, L=6e4)
    segs = cases.make_segments(model, seed=2)
    o = oracle.Oracle(model, 500, seed=2)
    o.init_prior(0.0)
    si = o.pack_segments(model, segs)
    for s in range(len(segs["start"])):
        o.update_segment(si, s)
        p = o.particles()
        assert abs(p["w_post"].sum() - 1) < 1e-12
        pos = min(segs["start"][s] + segs["length"][s], model["loci_length"])
        o.count(pos)
        if o.resample(pos):
            q = o.particles()
            assert np.allclose(q["w_pilot"], q["w_pilot"][0])     # pilot weights equalised (pc.cpp:350-351)
    _, parents = o.resample_events()
    for par in parents:
        assert (np.diff(par) >= 0).all() and par.min() >= 0 and par.max() < 500


def test_calibrated_survival_decreases_with_epoch_age(oracle):
    model = cases.make_model(n=4, E=8, L=1e7)
    med, trees = oracle.median_survival(model, seed=1, min_events=50, max_trees=32768)
    assert trees % 16384 == 0 and (med > 0).all()
    assert (np.diff(med[1:]) < 0).all()          # older nodes are hit sooner (larger branch length above them)


def test_focused_sampling_importance_weights_are_unbiased(oracle):
    """With -bias_heights/-bias_strengths recombinations are proposed 3x more often below 400 generations; the
    importance weights (particle.cpp:1106-1108) must undo that exactly: a no-data run still returns the model's
    rates and a likelihood of ~1 (E[weight] = 1)."""
    N0, rho = 1e4, 1e-8
    model = cases.make_model(n=4, E=1, N0=N0, rho=rho, L=4e5)
    model.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0], application_delays=[10000.0])
    segs = cases.nodata_segments(model)
    o = oracle.Oracle(model, 1500, seed=7)
    o.init_prior(0.0)
    o.run(o.pack_segments(model, segs))
    c = o.counts()
    assert abs(o.logl()) < 0.3
    assert abs(c["coal_count"][0] / c["coal_opp"][0] * 2 * N0 - 1) < 0.05
    assert abs(c["rec_count"][0] / c["rec_opp"][0] / rho - 1) < 0.05
    assert c["delayed_count"] > 0            # factors were pending (count.cpp:395-397)
    # more recombination events low in the tree than without focusing would give is not checked here: the
    # weighted rate above already proves proposal x weight = target


def test_delayed_factor_schedule(oracle):
    """DelayedFactor(final, f, cur, k=3) (particle.hpp:66-82): three applications of f^(1/3) at cur+d/7, +3d/7, +d.
    Checked through the invariant posterior = pilot * total_delayed on a run with data (particle.hpp:208)."""
    model = cases.make_model(n=4, E=8, L=6e4)
    model.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0], application_delays=np.array(model["lags"]) * 0.25)
    segs = cases.make_segments(model, seed=3)
    o = oracle.Oracle(model, 400, seed=3)
    o.init_prior(0.0)
    si = o.pack_segments(model, segs)
    for s in range(len(segs["start"])):
        o.update_segment(si, s)
        pos = min(segs["start"][s] + segs["length"][s], model["loci_length"])
        o.count(pos); o.resample(pos)
    p = o.particles()
    ratio = p["w_post"] / p["w_pilot"]       # = product of pending factors
    assert ratio.min() > 0 and (np.abs(np.log(ratio)) > 1e-3).any()


# ---------------------------------------------------------------- structured models (populations, migration, joins)

def _two_deme_model(n, sample_pops, mig=1.0, N0=1e4, E=1):
    m = cases.make_model(n=n, E=E, L=1e5)
    m = cases.make_structured(m, P=2, split_epoch=E + 5, mig=mig, sample_pops=sample_pops)   # no join: split beyond the last epoch
    m["single_mig"][:] = 0.0
    return m


def test_two_deme_coalescence_time_expectations(oracle):
    """Symmetric two-deme island model, per-lineage migration rate m = M/(4 N0): two lineages sampled in the same
    deme coalesce after 4N generations on average, two from different demes after 4N + 1/(2m) (Notohara 1990)."""
    N0 = 1e4
    for spops, expect in (([0, 0], 4 * N0), ([0, 1], 4 * N0 + 1.0 / (2 * 1.0 / (4 * N0)))):
        o = oracle.Oracle(_two_deme_model(2, spops), 40000, seed=17)
        o.init_prior(0.0)
        h = o.particles()["heights"][:, 0]
        assert abs(h.mean() / expect - 1) < 0.02, (spops, h.mean(), expect)
        mg = o.migrations()
        # every event list is sorted by time, sits on an existing branch below the root, and changes population
        for i in range(0, 40000, 997):
            k = mg["n_events"][i]
            t = mg["times"][i, :k]
            assert (np.diff(t) >= 0).all() and (t < h[i]).all()
            assert set(mg["branch"][i, :k].tolist()) <= {0, 1}


def test_structured_no_data_run_reproduces_model_rates(oracle):
    """Without data the filter samples the prior: coalescence, migration and recombination rates read back from
    the CountModel sums must equal the model's (cf. the acceptance ranges of test/old/newtests/test_two_pops.py)."""
    N0 = 1e4
    base = cases.make_model(n=4, E=6, L=1e6)
    model = cases.make_structured(base, P=2, split_epoch=4, mig=1.0)
    o = oracle.Oracle(model, 1000, seed=5)
    o.init_prior(0.0)
    o.run(o.pack_segments(model, cases.nodata_segments(model, 4000.0)))
    c = o.counts()
    coal = c["coal_count"] / np.maximum(c["coal_opp"], 1e-300) * 2 * N0
    ok = c["coal_count"] > 10
    assert ok.sum() >= 4 and np.abs(coal[ok] - 1).max() < 0.1
    mig = c["mig_count"].sum(2) / np.maximum(c["mig_opp"], 1e-300) * 4 * N0
    okm = c["mig_count"].sum(2) > 3
    assert okm.sum() >= 2 and np.abs(mig[okm] - 1).max() < 0.12
    assert c["coal_count"][4:, 1].sum() == 0 and c["mig_count"][4:].sum() == 0     # nobody is left in population 1
    assert abs(c["rec_count"].sum() / c["rec_opp"].sum() / 1e-8 - 1) < 0.03
    assert o.logl() == 0.0


def test_population_join_moves_every_lineage(oracle):
    """-ej: at the join every lineage of population 1 moves to population 0; nodes older than the join are in 0."""
    base = cases.make_model(n=6, E=8, L=1e5)
    model = cases.make_structured(base, P=2, split_epoch=4, mig=0.0)        # isolation until the join
    o = oracle.Oracle(model, 3000, seed=2)
    o.init_prior(0.0)
    p, mg = o.particles(), o.migrations()
    tj = model["change_times"][4]
    older = p["heights"] >= tj
    assert (mg["node_pops"][older] == 0).all()
    # without migration the only events are the joins, exactly at the boundary and into population 0
    k = mg["n_events"]
    assert k.max() <= 3 and k.sum() > 0
    for i in range(3000):
        assert (mg["times"][i, :k[i]] == tj).all() and (mg["newpop"][i, :k[i]] == 0).all()
    # lineages of different populations never coalesce before the join
    young = p["heights"] < tj
    ch = p["children"]
    spop = np.array(model["sample_pops"])
    for i in range(0, 3000, 37):
        for r in range(5):
            if young[i, r]:
                leaves = []
                stack = [int(ch[i, r, 0]), int(ch[i, r, 1])]
                while stack:
                    c_ = stack.pop()
                    if c_ < 6:
                        leaves.append(c_)
                    else:
                        stack += [int(ch[i, c_ - 6, 0]), int(ch[i, c_ - 6, 1])]
                assert len(set(spop[leaves])) == 1


def test_isolated_populations_cannot_coalesce(oracle):
    model = _two_deme_model(2, [0, 1], mig=0.0)
    o = oracle.Oracle(model, 4, seed=1)
    with pytest.raises(RuntimeError, match="No final coalescence"):
        o.init_prior(0.0)


def test_recombination_guide_importance_weights_are_unbiased(oracle):
    """A guide that is off from the model (sampling rates 0.7-1.4 x the true rate, uneven leaf rates), no data: the
    weighted counts still recover the model's recombination and coalescence rates and the likelihood stays at one
    (importance_weight_over_segment, particle.cpp:1159-1192; the event weights of samplePoint, particle.cpp:942-1010)."""
    n, E, L, K = 4, 6, 1e6, 8
    model = cases.make_model(n=n, E=E, L=L)
    rng = np.random.default_rng(1)
    leaf = rng.uniform(0.6, 1.4, (K, n))
    guide = dict(positions=np.arange(K) * L / K, rates=1e-8 * rng.uniform(0.7, 1.4, K), leaf_rates=leaf / leaf.sum(1, keepdims=True))
    segs = cases.nodata_segments(model, 4000.0)
    for extra in (dict(), dict(bias_heights=[400.0], bias_strengths=[3.0, 1.0])):
        m = dict(model, guide=guide, application_delays=np.full(E, 5000.0), **extra)
        o = oracle.Oracle(m, 1500, seed=3)
        o.init_prior(0.0)
        o.run(o.pack_segments(m, segs))
        c = o.counts()
        assert abs(c["rec_count"].sum() / c["rec_opp"].sum() / 1e-8 - 1) < 0.02
        assert np.abs(c["coal_count"][2:5] / c["coal_opp"][2:5] * 2e4 - 1).max() < 0.06
        assert abs(o.logl()) < 0.5


def test_focused_sampling_with_structure_is_unbiased(oracle):
    """Height-biased cut points in an isolation-with-migration model, no data: weighted counts still reproduce the
    model's recombination and per-population coalescence rates, and the likelihood stays at one."""
    n, E, L = 4, 6, 1e6
    model = cases.make_structured(cases.make_model(n=n, E=E, L=L), P=2)
    model = dict(model, bias_heights=[400.0], bias_strengths=[5.0, 1.0], application_delays=np.full(E, 5000.0))
    segs = cases.nodata_segments(model, 4000.0)
    o = oracle.Oracle(model, 1500, seed=3)
    o.init_prior(0.0)
    o.run(o.pack_segments(model, segs))
    c = o.counts()
    assert o.trace()["resampled"].sum() > 0                                     # the bias does move the weights
    assert abs(c["rec_count"].sum() / c["rec_opp"].sum() / 1e-8 - 1) < 0.02
    rates = c["coal_count"][2:4] / c["coal_opp"][2:4] * 2e4
    assert np.abs(rates - 1).max() < 0.06
    assert abs(o.logl()) < 0.5


@pytest.mark.parametrize("name", ["TestConstPopSize_MissingData", "TestConstPopSize_FourEpochs_MissingData"])
def test_oracle_meets_the_reference_no_data_bands(oracle, built_binary, name):
    """The oracle against numbers the reference itself holds: the two no-data regression classes
    (test/old/newtests/test_const_pop_size.py:150-170 and its four-epoch sibling; bands and flags in
    tests/golden/reference_bands.json).  All genotypes are missing, the filter samples the prior at Np = 100, and the bands
    are +-1 % around the truth -- they isolate the coalescent engine (reconstructed) from weighting and
    resampling.  As in tests/test_gpu_reference_bands.py::test_engine_reproduces_the_no_data_bands the recording limit far
    from data (smcsmc.cpp:266-275; the bands predate it) is lifted: mean over eight seeds inside every band, and at least
    six of the eight runs inside each."""
    import json
    import os
    import subprocess
    import reference_bands as rb
    from smcsmc_amd import segments as segmod
    c = [x for x in rb.load_cases() if x["name"] == name][0]
    argv = list(c["binary_argv"])
    seg = os.path.join(rb.GOLD, "seg", c["data"])
    argv[argv.index("@SEG@")] = seg
    out = subprocess.run([rb.BIN] + argv + ["-dumpmodel"], capture_output=True, text=True)
    m = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    E = len(m["change_times"])
    bh = [float(argv[argv.index("-bias_heights") + 1])]
    bs = [float(v) for v in argv[argv.index("-bias_strengths") + 1:argv.index("-bias_strengths") + 3]]
    lf = float(argv[argv.index("-calibrate_lag") + 1])
    model = dict(change_times=np.array(m["change_times"], float), pop_sizes=np.array(m["pop_sizes"], float)[:, 0], nsam=m["nsam"],
                 loci_length=float(m["loci_length"]), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"])
    model["lags"] = np.ones(E)
    med, _ = oracle.median_survival(model, seed=1, min_events=200, max_trees=1000000)
    model.update(lags=med * lf, bias_heights=bh, bias_strengths=bs, application_delays=med * 0.5, delay_type=0)
    S = segmod.Segments(seg, m["nsam"], m["loci_length"], max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    segs["max_record_epoch"][:] = E - 1
    N = model["pop_sizes"]
    est = []
    for seed in range(1, 9):
        o = oracle.Oracle(model, c["np"], seed=seed, max_trace_events=0)
        o.init_prior(segs["start"][0])
        o.run(o.pack_segments(model, segs))
        cn = o.counts()
        ne = (cn["coal_opp"] + 1.0) / (2 * (cn["coal_count"] + 1.0 / (2 * N)))          # with the prior pseudo-counts of the .out rows
        rec = (cn["rec_count"].sum() + E * model["recombination_rate"]) / (cn["rec_opp"].sum() + E)
        est.append(list(ne) + [rec])
        o.close()
    est = np.array(est)
    for t in c["targets"]:
        col = t["epoch"] if t["type"] == "Coal" else E
        vals = est[:, col]
        assert t["min"] <= vals.mean() <= t["max"], (rb.target_label(t), vals.mean(), t["min"], t["max"])
        if t["type"] != "Coal" or t["epoch"] > 0:
            assert ((vals >= t["min"]) & (vals <= t["max"])).sum() >= 6, (rb.target_label(t), vals)


def test_delayed_factor_store_capacity(oracle):
    """The reference's heap of delayed factors is unbounded (particle.hpp:248).  The restatement holds delay_cap of them per
    particle like the device path: with room nothing is forced; a full store is an error, or -- with delay_evict -- the
    earliest factor is applied early and counted."""
    model = cases.make_model(n=4, E=8, L=6e4)
    model.update(bias_heights=[400.0], bias_strengths=[8.0, 1.0], application_delays=np.array(model["lags"]) * 0.5)
    segs = cases.make_segments(cases.make_model(n=4, E=8, L=6e4), seed=27, max_seg_len=5000)

    def run(**kw):
        o = oracle.Oracle(model, 200, seed=3, **kw)
        o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
        return o
    o = run()
    st = o.delay_stats()
    assert st["forced"] == 0 and st["peak"] > 3
    logl = o.logl()
    o2 = run(delay_cap=max(2, st["peak"] // 3), delay_evict=True)
    st2 = o2.delay_stats()
    assert st2["forced"] > 0 and st2["peak"] == max(2, st["peak"] // 3)
    assert o2.logl() != logl            # early application changes the pilot weights, hence the resampling, hence everything
    with pytest.raises(RuntimeError, match="delayed-factor store overflow"):
        run(delay_cap=max(2, st["peak"] // 3))
