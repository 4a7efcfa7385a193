"""GPU parity for structured models (populations, migration, population joins): the HIP path against the
CPU oracle on identical seeded inputs, same bar as test_gpu_parity.py -- trees, migration lists, weights, ESS,
log-likelihood and resampling indices bit-identical; CountModel sums within 1e-9 relative."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

COUNT_RTOL = 1e-9


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _run_both(oracle, model, segs, Np, seed, ess=0.5, debug=0):
    from smcsmc_amd import ParticleFilter
    o = oracle.Oracle(model, Np, ess_fraction=ess, seed=seed, max_trace_events=64)
    o.init_prior(segs["start"][0])
    si = o.pack_segments(model, segs)
    g = ParticleFilter(model, Np, ess_fraction=ess, seed=seed, max_trace_events=64, debug=debug)
    g.init_prior(segs["start"][0])
    g.load_segments(segs)
    return o, si, g


def _canon_events(mg):
    """Events of every particle as a sorted list of (time, branch, newpop): events at the same instant on
    different branches (a population join crossed by two lineages) have no defined order."""
    out = []
    for i in range(len(mg["n_events"])):
        k = mg["n_events"][i]
        out.append(sorted(zip(_bits(mg["times"][i, :k]).tolist(), mg["branch"][i, :k].tolist(), mg["newpop"][i, :k].tolist())))
    return out


def _assert_state_equal(o, g):
    po, pg = o.particles(), g.particles()
    assert (po["children"] == pg["children"]).all()
    for k in ("heights", "w_post", "w_pilot", "next_base"):
        assert (_bits(po[k]) == _bits(pg[k])).all(), k
    mo, mg = o.migrations(), g.migrations()
    assert (mo["n_events"] == mg["n_events"]).all()
    assert (mo["node_pops"] == mg["node_pops"]).all()
    assert _canon_events(mo) == _canon_events(mg)


def _assert_counts_close(co, cg):
    for k in ("coal_count", "coal_opp", "coal_weight", "rec_count", "rec_opp", "rec_weight", "mig_count", "mig_opp",
              "mig_weight"):
        np.testing.assert_allclose(cg[k], co[k], rtol=COUNT_RTOL, atol=1e-300, err_msg=k)
    assert cg["resample_count"] == co["resample_count"]


def test_structured_prior_trees_bit_exact(oracle, hiplib):
    model = cases.make_structured(cases.make_model(n=6, E=8), P=2, split_epoch=5)
    segs = cases.nodata_segments(model)
    o, si, g = _run_both(oracle, model, segs, 777, seed=11)
    _assert_state_equal(o, g)
    assert o.migrations()["n_events"].sum() > 0


@pytest.mark.parametrize("debug", [0, 2, 1], ids=["register-tree", "register-tree-unfused", "lds-tree"])
@pytest.mark.parametrize("n,E,P,Np,seed", [(4, 8, 2, 600, 2), (8, 8, 2, 256, 3), (6, 6, 3, 300, 4), (2, 5, 2, 200, 6), (7, 12, 4, 320, 7)])
def test_structured_full_sweep_parity(oracle, hiplib, n, E, P, Np, seed, debug):
    """Both row kernels of the structured filter -- the register-resident tree (n <= 8, the default, completing the
    previous row while it loads; with PF_DEBUG_NO_FUSE as three launches per row) and the LDS tree (any n; forced here
    with PF_DEBUG_FORCE_LDS) -- against the oracle."""
    base = cases.make_model(n=n, E=E, L=1.0e5)
    segs = cases.make_segments(base, seed=seed, max_seg_len=5000)
    model = cases.make_structured(base, P=P, split_epoch=E - 3, mig=2.0)
    o, si, g = _run_both(oracle, model, segs, Np, seed, debug=debug)
    o.run(si)
    g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert g.segments_done() == len(to["T"])
    assert (to["resampled"] == tg["resampled"]).all()
    assert to["resampled"].sum() > 0, "test case must exercise resampling"
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg, pg_ = g.resample_events()
    assert (so == sg).all() and (po_ == pg_).all()
    _assert_state_equal(o, g)
    co, cg = o.counts(), g.counts()
    assert co["mig_count"].sum() > 0
    _assert_counts_close(co, cg)


def test_structured_asymmetric_sizes_and_sampling(oracle, hiplib):
    """Unequal population sizes, all samples from one population, one-way migration."""
    base = cases.make_model(n=4, E=6, L=8e4)
    segs = cases.make_segments(base, seed=5, max_seg_len=4000)
    model = cases.make_structured(base, P=2, split_epoch=4, mig=3.0, sizes=(1.0, 0.3), sample_pops=[0, 0, 0, 0])
    model["mig_rates"][:, 1, 0] = 0.0          # lineages enter population 1 backward in time and only return at the join
    o, si, g = _run_both(oracle, model, segs, 400, seed=9)
    o.run(si); g.run(); g.finish()
    assert (_bits(o.trace()["logl"]) == _bits(g.trace()["logl"])).all()
    _assert_state_equal(o, g)
    _assert_counts_close(o.counts(), g.counts())


def test_structured_prior_recovers_model(hiplib):
    """No-data run: the posterior is the prior, so the counts must reproduce the model's rates
    (the check of the reference's test_two_pops.py:76-119, here against known truth)."""
    from smcsmc_amd import ParticleFilter
    N0 = 1e4
    base = cases.make_model(n=4, E=8, L=4e6)
    model = cases.make_structured(base, P=2, split_epoch=5, mig=1.0)
    segs = cases.nodata_segments(model, 4000.0)
    g = ParticleFilter(model, 4096, seed=3); g.init_prior(0.0); g.load_segments(segs); g.run(); g.finish()
    c = g.counts()
    coal = c["coal_count"] / np.maximum(c["coal_opp"], 1e-300) * 2 * N0
    mig = c["mig_count"].sum(2) / np.maximum(c["mig_opp"], 1e-300) * 4 * N0
    # counts are posterior means per particle: 4096 independent prior ARGs stand behind every unit
    ok = c["coal_count"] > 20
    assert ok.sum() >= 6
    assert np.abs(coal[ok] - 1).max() < 0.05
    okm = c["mig_count"].sum(2) > 5
    assert okm.sum() >= 4
    assert np.abs(mig[okm] - 1).max() < 0.05
    assert c["coal_count"][5:, 1].sum() == 0 and c["mig_count"][5:].sum() == 0      # after the join
    assert abs(c["rec_count"].sum() / c["rec_opp"].sum() / 1e-8 - 1) < 0.02


def test_structured_lag_calibration_parity(oracle, hiplib):
    from smcsmc_amd import pf
    model = cases.make_structured(cases.make_model(n=4, E=8, L=1e7), P=2, split_epoch=5)
    dm, dt = pf.median_survival(model, seed=1, min_events=50, max_trees=32768)
    om, ot = oracle.median_survival(model, seed=1, min_events=50, max_trees=32768)
    assert dt == ot
    assert (_bits(dm) == _bits(om)).all()


def test_structured_without_a_way_to_coalesce_fails_loudly(hiplib):
    from smcsmc_amd import ParticleFilter, PfError
    model = cases.make_structured(cases.make_model(n=4, E=4, L=1e5), P=2, split_epoch=3, mig=0.0)
    model["single_mig"][:] = 0.0                   # two isolated populations for ever
    g = ParticleFilter(model, 128, seed=1)
    g.init_prior(0.0)
    with pytest.raises(PfError, match="No final coalescence"):
        g.sync()


def test_structured_binary_end_to_end(hiplib, tmp_path):
    """The drop-in binary on the reference's own two-population scrm data (first 2 Mb of
    test/old/newtests/testdata/twopopssplit_unidirmigr.seg: 4+4 haplotypes, split at 0.5, migration 0.2) with
    the kind of command line the front-end emits for test_two_pops.py:55-72.  Its .out must equal, character for
    character, the table written from the same run through the python binding, carry Coal rows per population
    and Migr rows, and the one-step estimates must sit near the simulation truth."""
    import json
    import os
    import subprocess
    from smcsmc_amd import ParticleFilter, outfile, segments as segmod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    if not os.path.exists(binary):
        from smcsmc_amd import build
        build.build_all()
    seg = os.path.join(root, "tests", "golden", "seg", "twopopssplit_unidirmigr_first2Mb.seg")
    L = 2000000
    core = ("-N0 10000 -t %g -r %g %d -I 2 4 4 -eN 0.0 1.0 -ema 0.0 0 0.2 0.2 0 -eN 0.1 1.0 -ema 0.1 0 0.2 0.2 0 "
            "-eN 0.5 1.0 -ema 0.5 0 0 0 0 -ej 0.5 2 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
    common = ["-nsam", "8", "-EM", "0", "-tmax", "4", "-seg", seg]
    r = subprocess.run([binary] + core + common + ["-Np", "1000", "-lag", "50000", "-seed", "5", "-o", str(tmp_path / "run")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(tmp_path / "run.out").read()
    m = json.loads(subprocess.run([binary] + core + ["-nsam", "8", "-tmax", "4", "-dumpmodel"], capture_output=True, text=True).stdout)
    E = len(m["change_times"])
    assert m["npop"] == 2 and m["sample_pops"] == [0, 0, 0, 0, 1, 1, 1, 1] and E == 3
    mig = np.array(m["mig_rates"]).reshape(E, 2, 2); smig = np.array(m["single_mig"]).reshape(E, 2, 2)
    assert mig[0, 0, 1] == pytest.approx(0.2 / 4e4) and mig[2].sum() == 0 and smig[2, 1, 0] == 1.0 and smig[:2].sum() == 0
    model = dict(change_times=np.array(m["change_times"]), pop_sizes=np.array(m["pop_sizes"]), lags=np.full(E, 50000.0),
                 nsam=8, loci_length=float(L), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"],
                 n_pops=2, mig_rates=mig, single_mig=smig, sample_pops=m["sample_pops"])
    S = segmod.Segments(seg, 8, L, max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    g = ParticleFilter(model, 1000, seed=5, max_trace_events=0)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    assert outfile.outfile_text(model, g.counts(), 1000) == text
    data = outfile.parse_outfile(text, is_text=True)
    assert data[(("LogL", -1, -1, -1, -1), "Count")] == pytest.approx(g.logl(), rel=1e-7)
    assert (("Migr", 0, 0, 1, -1), "Count") in data and (("Migr", 1, 1, 0, -1), "Opp") in data
    # one E-step from the truth stays near the truth (2 Mb of data: generous ranges)
    for key in [("Coal", 1, 0, -1, -1), ("Coal", 1, 1, -1, -1), ("Coal", 2, 0, -1, -1)]:
        ne = data[(key, "Opp")] / (2 * data[(key, "Count")])
        assert 6000 < ne < 16000, (key, ne)
    rec = data[(("Recomb", -1, -1, -1, -1), "Count")] / data[(("Recomb", -1, -1, -1, -1), "Opp")]
    assert 0.8e-8 < rec < 1.25e-8
    assert data[(("Migr", 2, 0, 1, -1), "Count")] == 0                      # after the join
    # default (calibrated) lags: the calibration kernel handles the structured model too
    r = subprocess.run([binary] + core + common + ["-Np", "200", "-seed", "3", "-o", str(tmp_path / "cal")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Migr" in open(tmp_path / "cal.out").read()
    # the regression configuration itself (test_two_pops.py:31-37): focused sampling, bias heights [400], strengths [10, 1]
    r = subprocess.run([binary] + core + common + ["-Np", "1000", "-seed", "8", "-bias_heights", "400", "-bias_strengths", "10", "1",
                                                   "-o", str(tmp_path / "bias")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Application delay for epoch" in r.stderr
    db = outfile.parse_outfile(str(tmp_path / "bias.out"))
    for key in [("Coal", 1, 0, -1, -1), ("Coal", 1, 1, -1, -1), ("Coal", 2, 0, -1, -1)]:
        ne = db[(key, "Opp")] / (2 * db[(key, "Count")])
        assert 6000 < ne < 16000, (key, ne)
    recb = db[(("Recomb", -1, -1, -1, -1), "Count")] / db[(("Recomb", -1, -1, -1, -1), "Opp")]
    assert 0.8e-8 < recb < 1.25e-8
    assert db[(("Delay", -1, -1, -1, -1), "Count")] > 0                      # importance weights were held back
    # auxiliary particle filter on the structured model: quantile tables from prior trees with migration
    r = subprocess.run([binary] + core + common + ["-Np", "500", "-seed", "2", "-lag", "50000", "-apf", "2", "-o", str(tmp_path / "apf")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Terminal branch length quantiles" in r.stdout
    da = outfile.parse_outfile(str(tmp_path / "apf.out"))
    ll_plain = data[(("LogL", -1, -1, -1, -1), "Count")]
    assert abs(da[(("LogL", -1, -1, -1, -1), "Count")] - ll_plain) < 0.02 * abs(ll_plain)
    # -arg on the structured model: R, C and M lines with their populations (pc.cpp:515-555)
    import gzip
    r = subprocess.run([binary] + core + common + ["-Np", "200", "-seed", "6", "-lag", "50000", "-arg", "-o", str(tmp_path / "arg")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw_lines = gzip.open(tmp_path / "arg.trees.gz", "rt").read().splitlines()
    import trees_format
    trees_format.check_lines(raw_lines, nsam=8, npop=2)   # the grammar of the reference's own example file (R, C and M lines)
    lines = [ln.split("\t") for ln in raw_lines]
    assert all(len(f) == 6 and f[0] in ("R", "C", "M") for f in lines)
    ms = [f for f in lines if f[0] == "M"]
    assert ms and all(f[3] in ("0", "1") and f[4] in ("0", "1") and f[3] != f[4] for f in ms)
    assert all(f[3] in ("0", "1") and f[4] == "-1" for f in lines if f[0] == "C") and all(f[3] == "-1" for f in lines if f[0] == "R")
    assert lines[-1][0] in ("C", "M") and float(lines[-1][1]) == 0.0                   # the first tree


@pytest.mark.parametrize("lds_tree", [False, True])
@pytest.mark.parametrize("n,P,delay_type", [(4, 2, 0), (8, 2, 0), (6, 3, 1), (4, 2, 2), (8, 2, 4)])
def test_focused_sampling_with_structure_parity(oracle, hiplib, n, P, delay_type, lds_tree):
    """-bias_heights / -bias_strengths with several populations (the configuration of the reference's own two-population
    regression tests, test_two_pops.py:36-37): biased cut point on the LDS tree, delayed importance weights, resampling
    with pending factors -- trees, migration events, weights, ESS and resampling indices bit-identical to the oracle."""
    from smcsmc_amd import ParticleFilter
    E = 6
    model = cases.make_structured(cases.make_model(n=n, E=E, L=1.2e5), P=P)
    model = dict(model, bias_heights=[400.0], bias_strengths=[5.0, 1.0], application_delays=np.full(E, 3000.0), delay_type=delay_type)
    segs = cases.make_segments(cases.make_model(n=n, E=E, L=1.2e5), seed=90 + n, max_seg_len=5000)
    o = oracle.Oracle(model, 320, seed=8, max_trace_events=64); o.init_prior(segs["start"][0]); si = o.pack_segments(model, segs)
    g = ParticleFilter(model, 320, seed=8, max_trace_events=64, debug=1 if lds_tree else 0)
    g.init_prior(segs["start"][0]); g.load_segments(segs)
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    bits = lambda a: np.asarray(a, dtype=np.float64).view(np.int64)   # noqa: E731
    for k in ("T", "ess", "logl"):
        assert (bits(to[k]) == bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg_, pg_ = g.resample_events()
    assert (so == sg_).all() and (po_ == pg_).all()
    po, pg = o.particles(), g.particles()
    assert (po["children"] == pg["children"]).all()
    for k in ("heights", "w_post", "w_pilot", "next_base"):
        assert (bits(po[k]) == bits(pg[k])).all(), k
    co, cg = o.counts(), g.counts()
    for k in ("coal_count", "coal_opp", "rec_count", "rec_opp", "mig_count", "mig_opp"):       # sums in another order
        np.testing.assert_allclose(cg[k], co[k], rtol=1e-9, atol=1e-9 * np.abs(co[k]).max(), err_msg=k)
    np.testing.assert_allclose(cg["delayed_opp"], co["delayed_opp"], rtol=1e-12)


def test_terminal_branch_quantiles_with_structure(oracle, hiplib):
    """calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166) from prior trees of an isolation-with-migration
    model: device == oracle bit for bit."""
    from smcsmc_amd import pf
    model = cases.make_structured(cases.make_model(n=6, E=8, L=1e6), P=2)
    dl, dm = pf.terminal_branch_quantiles(model, seed=1, n_trees=30000)
    ol_, om = oracle.terminal_branch_quantiles(model, seed=1, n_trees=30000)
    bits = lambda a: np.asarray(a, dtype=np.float64).view(np.int64)   # noqa: E731
    assert (bits(dl) == bits(ol_)).all() and bits([dm])[0] == bits([om])[0]
    assert (np.diff(dl, axis=1) > 0).all()
    # structure lengthens the trees: more than the panmictic expectation 4N H(n-1)
    assert dm > 4e4 * sum(1.0 / k for k in range(1, 6))


@pytest.mark.parametrize("n,P,level", [(4, 2, 2), (8, 2, 3)])
def test_auxiliary_particle_filter_with_structure_parity(oracle, hiplib, n, P, level):
    """-apf with several populations: the look-ahead factor (particle.cpp:439-617) depends on the local tree only; the
    quantile tables come from prior trees of the structured model.  Trees, weights, ESS, resampling indices as the oracle's."""
    from smcsmc_amd import ParticleFilter, pf, segments as segmod
    E = 6
    base = cases.make_model(n=n, E=E, L=1.2e5)
    model = cases.make_structured(base, P=P)
    segs = cases.make_segments(base, seed=30 + n, max_seg_len=5000)
    rows = [(int(s) + 1, int(l), int(st), list(map(int, a)))
            for s, l, st, a in zip(segs["start"], segs["length"], segs["state"], segs["alleles"])]
    la = segmod.pack_lookahead(rows, n)
    tbl = pf.terminal_branch_quantiles(model, seed=1, n_trees=20000)
    o = oracle.Oracle(model, 320, seed=9, max_trace_events=64); o.init_prior(segs["start"][0]); si = o.pack_segments(model, segs)
    g = ParticleFilter(model, 320, seed=9, max_trace_events=64); g.init_prior(segs["start"][0]); g.load_segments(segs)
    o.load_lookahead(la, level, tbl); g.load_lookahead(la, level, tbl)
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    bits = lambda a: np.asarray(a, dtype=np.float64).view(np.int64)   # noqa: E731
    for k in ("T", "ess", "logl"):
        assert (bits(to[k]) == bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg_, pg_ = g.resample_events()
    assert (so == sg_).all() and (po_ == pg_).all()
    po, pg = o.particles(), g.particles()
    assert (po["children"] == pg["children"]).all()
    for k in ("heights", "w_post", "w_pilot", "next_base"):
        assert (bits(po[k]) == bits(pg[k])).all(), k
    assert not np.allclose(pg["w_post"], pg["w_pilot"])          # the look-ahead sits in the pilot weight only


@pytest.mark.parametrize("lds_tree", [False, True])
@pytest.mark.parametrize("n,P,bias", [(4, 2, False), (8, 2, True)])
def test_recombination_guide_with_structure_parity(oracle, hiplib, n, P, bias, lds_tree):
    """-guide with several populations: position-dependent sampling rate, per-sample relative rates, importance weights
    over the stretch and per event (particle.cpp:942-1254), alone and together with the height bias."""
    from smcsmc_amd import ParticleFilter
    E = 6
    base = cases.make_model(n=n, E=E, L=1.2e5)
    rng = np.random.default_rng(n)
    K = 9
    leaf = rng.uniform(0.4, 2.5, (K, n)); leaf /= leaf.sum(1, keepdims=True)
    guide = dict(positions=np.floor(np.arange(K) * 1.2e5 / K), rates=1e-8 * rng.uniform(0.4, 2.5, K), leaf_rates=leaf)
    model = dict(cases.make_structured(base, P=P), guide=guide, application_delays=np.full(E, 3000.0))
    if bias:
        model.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0])
    segs = cases.make_segments(base, seed=60 + n, max_seg_len=5000)
    o = oracle.Oracle(model, 320, seed=6, max_trace_events=64); o.init_prior(segs["start"][0]); si = o.pack_segments(model, segs)
    g = ParticleFilter(model, 320, seed=6, max_trace_events=64, debug=1 if lds_tree else 0)
    g.init_prior(segs["start"][0]); g.load_segments(segs)
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    bits = lambda a: np.asarray(a, dtype=np.float64).view(np.int64)   # noqa: E731
    for k in ("T", "ess", "logl"):
        assert (bits(to[k]) == bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg_, pg_ = g.resample_events()
    assert (so == sg_).all() and (po_ == pg_).all()
    po, pg = o.particles(), g.particles()
    assert (po["children"] == pg["children"]).all()
    for k in ("heights", "w_post", "w_pilot", "next_base"):
        assert (bits(po[k]) == bits(pg[k])).all(), k
    co, cg = o.counts(), g.counts()
    for k in ("coal_count", "coal_opp", "rec_count", "rec_opp", "mig_count", "mig_opp"):
        np.testing.assert_allclose(cg[k], co[k], rtol=1e-9, atol=1e-9 * np.abs(co[k]).max(), err_msg=k)


@pytest.mark.parametrize("n,P,Np,bias", [(4, 2, 300, False), (8, 2, 160, False), (6, 3, 200, True)])
def test_tree_dump_with_structure(oracle, hiplib, n, P, Np, bias):
    """-arg with several populations (pc.cpp:515-555, particle.cpp:292-298): the R, C and M lines of the drawn particle's
    history -- positions, heights, populations and descendants -- equal the oracle's linked list event for event; every
    update reads R, C, then the migrations of the walk latest first, and each migration leaves the population the
    next one (earlier in the file order: later in time) starts from."""
    from smcsmc_amd import ParticleFilter, outfile
    E = 8
    base = cases.make_model(n=n, E=E, L=1.2e5)
    segs = cases.make_segments(base, seed=40 + n, max_seg_len=5000)
    model = cases.make_structured(base, P=P, split_epoch=E - 3, mig=2.0)
    if bias:
        model = dict(model, bias_heights=[400.0], bias_strengths=[4.0, 1.0], application_delays=np.full(E, 3000.0))
    g = ParticleFilter(model, Np, seed=5, max_trace_events=0, record_trees=True, log_cap=8192, gen_cap=4096)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    assert g.trace()["resampled"].sum() > 3
    part, kind, pos, hgt, desc, fr, to = g.sample_tree_events(pops=True)
    o = oracle.Oracle(model, Np, seed=5, max_trace_events=0)
    o.enable_tree_recording()
    o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
    opart, okind, opos, ohgt, odesc, ofr, oto = o.sample_tree_events(pops=True)
    assert opart == part and len(okind) == len(kind)
    assert (okind == kind).all() and (odesc == desc).all() and (ofr == fr).all() and (oto == to).all()
    assert (_bits(opos) == _bits(pos)).all() and (_bits(ohgt) == _bits(hgt)).all()
    assert (kind == 2).sum() > 0, "the case must exercise migrations"
    assert (np.diff(pos) <= 0).all()
    full = (1 << n) - 1
    i = 0
    while i < len(kind):
        if kind[i] == 0:                       # an update: R, C, M...
            cut = int(desc[i]); x = pos[i]; i += 1
            assert kind[i] == 1 and pos[i] == x and hgt[i] >= hgt[i - 1] and (int(desc[i]) & cut) == cut
        else:                                  # a leaf of the first tree: C, M...
            assert kind[i] == 1 and pos[i] == 0.0
            cut = None
        tc = hgt[i]; i += 1
        while i < len(kind) and kind[i] == 2:
            assert hgt[i] <= tc and fr[i] != to[i] and 0 <= to[i] < P
            if cut is not None:
                assert int(desc[i]) in (cut, full & ~cut)
            i += 1
    text = outfile.trees_text(kind, pos, hgt, desc, start_position=1.0, from_pop=fr, to_pop=to)
    assert any(ln.startswith("M\t") for ln in text.splitlines()) and len(text.splitlines()) == len(kind)


def test_migration_event_capacity_is_a_parameter(oracle, hiplib):
    """pf_params.mig_cap: a model with fast migration and no join puts more than the default 96 migration events on a
    local tree at some point of the sweep.  With the default the run stops with a reported error (on the device and in the
    oracle alike); with room for 160 it runs through and matches the oracle, event lists included."""
    from smcsmc_amd import ParticleFilter, PfError
    E, n = 5, 6
    base = cases.make_model(n=n, E=E, L=6e4)
    segs = cases.make_segments(base, seed=77, max_seg_len=5000)
    model = cases.make_structured(base, P=2, split_epoch=E, mig=5.0)
    with pytest.raises(RuntimeError, match="too many migration events"):
        o = oracle.Oracle(model, 200, seed=3, max_trace_events=8)
        o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
    with pytest.raises(PfError, match="too many migration events"):
        g = ParticleFilter(model, 200, seed=3, max_trace_events=8)
        g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    for debug in (0, 1):
        o = oracle.Oracle(model, 200, seed=3, max_trace_events=8, mig_cap=160)
        o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
        g = ParticleFilter(model, 200, seed=3, max_trace_events=8, mig_cap=160, debug=debug)
        g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
        assert (_bits(o.trace()["logl"]) == _bits(g.trace()["logl"])).all()
        _assert_state_equal(o, g)
        _assert_counts_close(o.counts(), g.counts())
    with pytest.raises(PfError, match="does not fit the LDS"):
        ParticleFilter(model, 200, seed=3, mig_cap=600)
