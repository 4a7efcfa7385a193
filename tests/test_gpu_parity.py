"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): resampling indices bit-exact; log-likelihood / weight sums within
1e-6 relative.  What is actually asserted is stronger: weights, ESS, log-likelihood and tree heights
are bit-identical (both sides execute the same IEEE-754 operations in the same order); only the
CountModel sums, accumulated in a different order by design, use a relative tolerance of 1e-9.
"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

COUNT_RTOL = 1e-9


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_device_math_bit_exact(oracle, hiplib):
    from smcsmc_amd import pf
    L = oracle.lib()
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-50, 5, 20000), rng.uniform(-0.8, 0.8, 20000), -rng.exponential(1e-6, 2000),
                        np.array([0.0, -0.0, -745.0, -800.0, 1e-300, 700.0])])
    e, l, f = pf.device_math(x)
    eo = np.array([L.smco_exp(v) for v in x]); fo = np.array([L.smco_fastexp(v) for v in x])
    assert (_bits(e) == _bits(eo)).all()
    assert (_bits(f) == _bits(fo)).all()
    pos = np.abs(x[x != 0]) * rng.uniform(1e-12, 1e3, (x != 0).sum())
    pos = np.concatenate([pos, rng.uniform(0, 1, 20000), [5e-324, 1e-310, 1.0, 0.999999999999]])
    _, lg, _ = pf.device_math(pos)
    lo = np.array([L.smco_log(v) for v in pos])
    assert (_bits(lg) == _bits(lo)).all()
    # and the portable exp/log are accurate
    assert np.max(np.abs(eo[:40000] / np.exp(x[:40000]) - 1)) < 4e-16
    assert np.max(np.abs(lo - np.log(pos))[np.abs(np.log(pos)) > 1e-3] / np.abs(np.log(pos))[np.abs(np.log(pos)) > 1e-3]) < 4e-16


def test_device_division_is_ieee(hiplib):
    from smcsmc_amd import pf
    rng = np.random.default_rng(7)
    a = rng.standard_normal(200000) * 10.0 ** rng.integers(-200, 200, 200000)
    b = rng.standard_normal(200000) * 10.0 ** rng.integers(-100, 100, 200000)
    assert (_bits(pf.device_div(a, b)) == _bits(a / b)).all()


def test_device_philox_bit_exact(oracle, hiplib):
    from smcsmc_amd import pf
    L = oracle.lib()
    for seed, slot, stream, first in [(1, 0, 0, 0), (12345678901234, 9999, 0, 2**33), (7, 0xFFFFFFFF, 1, 5)]:
        d = pf.device_uniform(seed, slot, stream, first, 4096)
        o = np.array([L.smco_uniform(seed, slot, stream, first + i) for i in range(4096)])
        assert (_bits(d) == _bits(o)).all()
        assert d.min() > 0 and d.max() < 1


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097, 10000, 70001])
def test_canonical_reduction_and_scan(oracle, hiplib, n):
    from smcsmc_amd import pf
    L = oracle.lib()
    rng = np.random.default_rng(n)
    x = rng.exponential(1.0, n) * 10.0 ** rng.integers(-30, 0, n)
    s, sc = pf.device_reduce(x)
    so = L.smco_canon_sum(x.ctypes.data, n)
    sco = np.zeros(n); L.smco_canon_scan(x.ctypes.data, sco.ctypes.data, n)
    assert _bits([s])[0] == _bits([so])[0]
    assert (_bits(sc) == _bits(sco)).all()


@pytest.mark.parametrize("n", [2, 64, 100, 1000, 10000])
def test_systematic_resampling_offsets_bit_exact(oracle, hiplib, n):
    from smcsmc_amd import pf
    L = oracle.lib()
    rng = np.random.default_rng(100 + n)
    for trial in range(3):
        w = rng.exponential(1.0, n)
        if trial == 1:
            w[rng.integers(0, n, n // 2)] *= 1e-40          # many (near-)empty particles
        if trial == 2:
            w[:] = 1e-300; w[n // 2] = 1.0                  # one particle takes everything
        lo = pf.device_systematic(w)
        u = L.smco_uniform(1, 0xFFFFFFFF, 1, 0)
        ref = np.zeros(n + 1, np.int32)
        L.smco_systematic(w.ctypes.data, n, u, ref.ctypes.data)
        assert (lo == ref).all()
        assert lo[0] == 0 and lo[-1] == n and (np.diff(lo) >= 0).all()


def _run_both(oracle, model, segs, Np, seed, ess=0.5, **kw):
    from smcsmc_amd import ParticleFilter
    o = oracle.Oracle(model, Np, ess_fraction=ess, seed=seed, max_trace_events=64, **kw)
    o.init_prior(segs["start"][0])
    si = o.pack_segments(model, segs)
    g = ParticleFilter(model, Np, ess_fraction=ess, seed=seed, max_trace_events=64, **kw)
    g.init_prior(segs["start"][0])
    g.load_segments(segs)
    return o, si, g


def _assert_state_equal(o, g):
    po, pg = o.particles(), g.particles()
    assert (po["children"] == pg["children"]).all()
    for k in ("heights", "w_post", "w_pilot", "next_base"):
        assert (_bits(po[k]) == _bits(pg[k])).all(), k


def _assert_counts_close(co, cg):
    for k in ("coal_count", "coal_opp", "coal_weight", "rec_count", "rec_opp", "rec_weight"):
        np.testing.assert_allclose(cg[k], co[k], rtol=COUNT_RTOL, atol=1e-300, err_msg=k)
    assert cg["resample_count"] == co["resample_count"]
    np.testing.assert_allclose(cg["delayed_opp"], co["delayed_opp"], rtol=1e-12)


def test_prior_initial_trees_bit_exact(oracle, hiplib):
    model = cases.make_model(n=6, E=8)
    segs = cases.nodata_segments(model)
    o, si, g = _run_both(oracle, model, segs, 777, seed=11)
    _assert_state_equal(o, g)


@pytest.mark.parametrize("n,E,Np,seed", [(2, 1, 300, 1), (4, 8, 1000, 2), (8, 16, 256, 3), (3, 4, 65, 4), (6, 64, 200, 5), (4, 33, 130, 6), (12, 8, 128, 7)])
def test_full_sweep_parity(oracle, hiplib, n, E, Np, seed):
    model = cases.make_model(n=n, E=E, L=1.5e5)
    segs = cases.make_segments(model, seed=seed, max_seg_len=5000)
    o, si, g = _run_both(oracle, model, segs, Np, seed)
    o.run(si)
    g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    m = len(to["T"])
    assert g.segments_done() == m
    assert (to["resampled"] == tg["resampled"]).all()
    assert to["resampled"].sum() > 0, "test case must exercise resampling"
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg, pg_ = g.resample_events()
    assert (so == sg).all() and (po_ == pg_).all()          # resampling indices bit-exact
    _assert_state_equal(o, g)
    assert _bits([o.logl()])[0] == _bits([g.logl()])[0]
    _assert_counts_close(o.counts(), g.counts())


def test_stepwise_api_matches_run(oracle, hiplib):
    from smcsmc_amd import ParticleFilter
    model = cases.make_model(n=4, E=4, L=5e4)
    segs = cases.make_segments(model, seed=9)
    a = ParticleFilter(model, 500, seed=5); a.init_prior(0.0); a.load_segments(segs); a.run(); a.finish()
    b = ParticleFilter(model, 500, seed=5); b.init_prior(0.0); b.load_segments(segs)
    for s in range(len(segs["start"])):
        b.update_segment(s); b.count(s); b.resample(s)
    b.finish()
    assert _bits([a.logl()])[0] == _bits([b.logl()])[0]
    ca, cb = a.counts(), b.counts()
    for k in ("coal_count", "coal_opp", "rec_count", "rec_opp"):
        assert (_bits(ca[k]) == _bits(cb[k])).all()


def test_unphased_missing_and_partial_segments(oracle, hiplib):
    model = cases.make_model(n=4, E=8, L=1.2e5, dephase=False)
    segs = cases.make_segments(model, seed=21, unphased=True, missing_block=(30000, 60000, (2, 3)), max_seg_len=2000)
    o, si, g = _run_both(oracle, model, segs, 600, seed=8)
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (_bits(to["logl"]) == _bits(tg["logl"])).all()
    assert (to["resampled"] == tg["resampled"]).all()
    _assert_state_equal(o, g)
    _assert_counts_close(o.counts(), g.counts())


def test_all_missing_gap_limits_recording(oracle, hiplib):
    """A long all-missing stretch: max_epoch_to_update (smcsmc.cpp:266-275) switches recording off."""
    model = cases.make_model(n=4, E=8, L=3e5)
    segs = cases.make_segments(model, seed=33, missing_block=(100000, 220000, (0, 1, 2, 3)), max_seg_len=5000)
    assert segs["max_record_epoch"].min() < len(model["lags"]) - 1
    o, si, g = _run_both(oracle, model, segs, 400, seed=3)
    o.run(si); g.run(); g.finish()
    assert (_bits(o.trace()["logl"]) == _bits(g.trace()["logl"])).all()
    _assert_counts_close(o.counts(), g.counts())


def test_determinism_run_to_run(hiplib):
    from smcsmc_amd import ParticleFilter
    model = cases.make_model(n=4, E=8, L=8e4)
    segs = cases.make_segments(model, seed=2)
    out = []
    for _ in range(2):
        g = ParticleFilter(model, 1000, seed=42); g.init_prior(0.0); g.load_segments(segs); g.run(); g.finish()
        c = g.counts(); out.append(np.concatenate([c[k] for k in ("coal_count", "coal_opp", "rec_count", "rec_opp")]))
        g.close()
    assert (_bits(out[0]) == _bits(out[1])).all()


def test_ancestral_aware_flag(oracle, hiplib):
    model = cases.make_model(n=4, E=4, L=6e4, ancestral_aware=True)
    segs = cases.make_segments(model, seed=4)
    o, si, g = _run_both(oracle, model, segs, 300, seed=6)
    o.run(si); g.run(); g.finish()
    assert _bits([o.logl()])[0] == _bits([g.logl()])[0]


def test_lag_calibration_parity(oracle, hiplib):
    """calculate_median_survival_distances (smcsmc.cpp:169-263): device batches == oracle batches, bit for bit."""
    from smcsmc_amd import pf
    model = cases.make_model(n=4, E=8, L=1e7)
    dm, dt = pf.median_survival(model, seed=1, min_events=50, max_trees=32768)
    om, ot = oracle.median_survival(model, seed=1, min_events=50, max_trees=32768)
    assert dt == ot
    assert (_bits(dm) == _bits(om)).all()
    lags = pf.calibrated_lags(model, lag_fraction=2.0)
    assert (lags > 0).all() and lags[1] > lags[-1]


def test_binary_end_to_end_matches_library_and_front_end_contract(oracle, hiplib, tmp_path):
    """The drop-in binary on the reference's scrm data: its .out must equal, character for character, the table
    written from the same run through the python binding, and parse with the front-end's reader."""
    import json
    import os
    import subprocess
    from smcsmc_amd import ParticleFilter, outfile, segments as segmod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    if not os.path.exists(binary):
        from smcsmc_amd import build
        build.build_all()
    seg = os.path.join(root, "tests", "golden", "seg", "constpopsize_first3000.seg")
    L = 2000000
    core = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.01 1 -eN 0.25 1 -eN 0.5 1 -eN 1 1 -eN 1.5 1"
            % (4 * 1e4 * 2.5e-8 * L, 4 * 1e4 * 1e-8 * L, L)).split()
    args = core + ["-nsam", "2", "-Np", "400", "-EM", "0", "-tmax", "4", "-lag", "20000", "-seed", "7", "-seg", seg,
                   "-record_ess", "-o", str(tmp_path / "run")]
    r = subprocess.run([binary] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(tmp_path / "run.out").read()
    assert os.path.exists(tmp_path / "run.log") and os.path.exists(tmp_path / "run.recomb.gz") and os.path.exists(tmp_path / "run.resample")
    m = json.loads(subprocess.run([binary] + core + ["-nsam", "2", "-tmax", "4", "-dumpmodel"], capture_output=True, text=True).stdout)
    E = len(m["change_times"])
    model = dict(change_times=np.array(m["change_times"]), pop_sizes=np.array(m["pop_sizes"])[:, 0], lags=np.full(E, 20000.0),
                 nsam=2, loci_length=float(L), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"])
    S = segmod.Segments(seg, 2, L, max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    g = ParticleFilter(model, 400, seed=7, max_trace_events=0)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    assert outfile.outfile_text(model, g.counts(), 400) == text
    data = outfile.parse_outfile(text, is_text=True)
    assert data[(("LogL", -1, -1, -1, -1), "Count")] == pytest.approx(g.logl(), rel=1e-7)
    # and the oracle agrees with what the binary wrote
    o = oracle.Oracle(model, 400, seed=7); o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
    assert _bits([o.logl()])[0] == _bits([g.logl()])[0]
    np.testing.assert_allclose(g.counts()["coal_count"], o.counts()["coal_count"], rtol=COUNT_RTOL)
    # default (calibrated) lags also run
    r = subprocess.run([binary] + core + ["-nsam", "2", "-Np", "200", "-EM", "0", "-tmax", "4", "-seed", "3", "-seg", seg,
                                          "-o", str(tmp_path / "cal")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "LogL" in open(tmp_path / "cal.out").read()
    # the reference's regression configuration: focused sampling with delayed importance weights
    # (test/old/newtests/test_const_pop_size.py:29-37: lag 2, Np 1000, bias_heights [400], bias_strengths [3,1], tmax 4)
    r = subprocess.run([binary] + core + ["-nsam", "2", "-Np", "1000", "-EM", "0", "-tmax", "4", "-calibrate_lag", "2", "-seed", "1",
                                          "-bias_heights", "400", "-bias_strengths", "3", "1", "-seg", seg,
                                          "-o", str(tmp_path / "bias")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = outfile.parse_outfile(str(tmp_path / "bias.out"))
    rec = d[(("Recomb", -1, -1, -1, -1), "Count")] / d[(("Recomb", -1, -1, -1, -1), "Opp")]
    assert 0.8e-8 < rec < 1.2e-8                      # truth 1e-8
    assert d[(("Delay", -1, -1, -1, -1), "Count")] > 0


@pytest.mark.parametrize("n,delay_type", [(4, 0), (2, 1), (6, 0), (12, 0), (10, 1), (4, 4), (12, 5)])   # + 4: every factor delayed
def test_focused_sampling_and_delayed_importance_weights(oracle, hiplib, n, delay_type):
    """-bias_heights / -bias_strengths with delayed application of the importance weights
    (particle.cpp:866-891, 1020-1126; particle.hpp:59-101, 185-209): the configuration of the reference's own
    regression tests (test_const_pop_size.py:34-35: bias_heights [400], bias_strengths [3, 1])."""
    model = cases.make_model(n=n, E=8, L=1.2e5)
    model.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0], delay_type=delay_type,
                 application_delays=np.array(model["lags"]) * 0.25)
    segs = cases.make_segments(model, seed=17 + n, max_seg_len=5000)
    o, si, g = _run_both(oracle, model, segs, 700 if n <= 8 else 200, seed=5)      # n > 8: the LDS-tree kernel
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg, pg_ = g.resample_events()
    assert (so == sg).all() and (po_ == pg_).all()
    _assert_state_equal(o, g)
    co, cg = o.counts(), g.counts()
    _assert_counts_close(co, cg)
    assert co["delayed_count"] > 0 and cg["delayed_count"] == pytest.approx(co["delayed_count"], rel=1e-12)
    # pilot and posterior weights differ while factors are pending
    p = g.particles()
    assert not np.allclose(p["w_post"], p["w_pilot"])


@pytest.mark.parametrize("n,P", [(4, 1), (12, 1), (4, 2), (10, 2)])    # register-tree and LDS-tree kernels, one and two populations
def test_delayed_factor_store_is_bounded_and_says_so(oracle, hiplib, n, P):
    """The reference keeps a particle's delayed importance factors in an unbounded heap (particle.hpp:59-101, 248); the
    device keeps pf_params.delay_cap of them per particle.  A full store stops the run with a message, as every other
    bounded ring does; with delay_evict the earliest factor is applied ahead of its position instead, and both sides count
    how often (same number, same weights).  With room enough nothing is ever forced."""
    from smcsmc_amd.pf import PfError
    model = cases.make_model(n=n, E=8, L=1.0e5)
    model.update(bias_heights=[400.0], bias_strengths=[8.0, 1.0], application_delays=np.array(model["lags"]) * 0.5)
    if P > 1:
        model = cases.make_structured(model, P=P, split_epoch=5, mig=1.0)
    segs = cases.make_segments(cases.make_model(n=n, E=8, L=1.0e5), seed=23 + n, max_seg_len=5000)
    Np = 300 if n <= 8 else 128
    # (1) plenty of room: the peak is reported, nothing is forced
    o, si, g = _run_both(oracle, model, segs, Np, seed=3)
    o.run(si); g.run(); g.finish()
    so, sg = o.delay_stats(), g.delay_stats()
    assert so == sg and so["forced"] == 0 and 3 < so["peak"] <= 128, (so, sg)
    ref_logl = g.logl()
    cap = max(2, so["peak"] // 3)
    # (2) a store a third that size, eviction allowed: counted, identical on both sides, and the weights do change
    o, si, g = _run_both(oracle, model, segs, Np, seed=3, delay_cap=cap, delay_evict=True)
    o.run(si); g.run(); g.finish()
    so, sg = o.delay_stats(), g.delay_stats()
    assert so == sg and so["forced"] > 0 and so["peak"] == cap, (so, sg)
    assert (_bits(o.trace()["logl"]) == _bits(g.trace()["logl"])).all()
    _assert_state_equal(o, g)
    assert g.trace()["ess"].tolist() != [] and g.logl() != ref_logl or so["forced"] > 0
    # (3) the same store without the switch: an error on both sides
    o, si, g = _run_both(oracle, model, segs, Np, seed=3, delay_cap=cap)
    with pytest.raises(RuntimeError, match="delayed-factor store overflow"):
        o.run(si)
    with pytest.raises(PfError, match="delayed-factor store overflow"):
        g.run(); g.finish()


def test_three_bias_bands(oracle, hiplib):
    model = cases.make_model(n=4, E=6, L=8e4)
    model.update(bias_heights=[300.0, 5000.0], bias_strengths=[4.0, 1.0, 0.5], application_delays=np.full(6, 3000.0))
    segs = cases.make_segments(model, seed=44, max_seg_len=5000)
    o, si, g = _run_both(oracle, model, segs, 300, seed=9)
    o.run(si); g.run(); g.finish()
    assert (_bits(o.trace()["logl"]) == _bits(g.trace()["logl"])).all()
    _assert_counts_close(o.counts(), g.counts())


# ---------------------------------------------------------------- auxiliary particle filter (-apf)

def _lookahead_inputs(model, segs):
    from smcsmc_amd import segments as segmod
    n = model["nsam"]
    rows = [(int(s) + 1, int(l), int(st), list(map(int, a)))
            for s, l, st, a in zip(segs["start"], segs["length"], segs["state"], segs["alleles"])]
    return segmod.pack_lookahead(rows, n)


def test_terminal_branch_length_quantiles_parity(oracle, hiplib):
    """calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166): device == oracle bit for bit, and the mean
    total branch length is the coalescent expectation 4N * H(n-1)."""
    from smcsmc_amd import pf
    model = cases.make_model(n=6, E=8, L=1e6)
    dl, dm = pf.terminal_branch_quantiles(model, seed=1, n_trees=60000)
    ol_, om = oracle.terminal_branch_quantiles(model, seed=1, n_trees=60000)
    assert (_bits(dl) == _bits(ol_)).all() and _bits([dm])[0] == _bits([om])[0]
    expect = 4e4 * sum(1.0 / k for k in range(1, 6))
    assert abs(dm / expect - 1) < 0.01
    assert (np.diff(dl, axis=1) > 0).all()


@pytest.mark.parametrize("n,level,unphased", [(4, 1, False), (4, 2, True), (8, 3, False), (8, 4, True)])
def test_auxiliary_particle_filter_parity(oracle, hiplib, n, level, unphased):
    """update_lookahead_likelihood / includeLookaheadLikelihood (pc.cpp:227-240, particle.cpp:439-617): the look-ahead
    factor enters the pilot weight only; trees, weights, ESS and resampling indices stay bit-identical to the oracle."""
    from smcsmc_amd import pf
    model = cases.make_model(n=n, E=8, L=1.2e5)
    segs = cases.make_segments(model, seed=20 + n + level, unphased=unphased, max_seg_len=5000,
                               missing_block=(40000, 60000, (0, 1)) if level == 2 else None)
    la = _lookahead_inputs(model, segs)
    tbl = pf.terminal_branch_quantiles(model, seed=1, n_trees=20000)
    assert (la["n_doubletons"] > 0).any()
    if level >= 3:
        assert (la["first_split_distance"] > 0).any()
    o, si, g = _run_both(oracle, model, segs, 400, seed=9)
    o.load_lookahead(la, level, tbl); g.load_lookahead(la, level, tbl)
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg_, pg_ = g.resample_events()
    assert (so == sg_).all() and (po_ == pg_).all()
    _assert_state_equal(o, g)
    _assert_counts_close(o.counts(), g.counts())
    # the look-ahead changes which particles survive, not the estimator: same data without it gives another run
    o0, si0, g0 = _run_both(oracle, model, segs, 400, seed=9)
    g0.run(); g0.finish()
    assert (g0.trace()["ess"] != tg["ess"]).any()
    assert abs(g0.logl() - g.logl()) < 0.05 * abs(g.logl())


def test_binary_auxiliary_particle_filter(hiplib, tmp_path):
    """bin/smcsmc -apf 2 equals the library run with the same look-ahead tables, character for character."""
    import json
    import os
    import subprocess
    from smcsmc_amd import ParticleFilter, outfile, pf, segments as segmod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    seg = os.path.join(root, "tests", "golden", "seg", "constpopsize_first3000.seg")
    L = 1000000
    core = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.01 1 -eN 0.25 1 -eN 1 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
    r = subprocess.run([binary] + core + ["-nsam", "2", "-Np", "300", "-EM", "0", "-tmax", "4", "-lag", "20000", "-seed", "4",
                                          "-apf", "2", "-seg", seg, "-o", str(tmp_path / "apf")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Terminal branch length quantiles" in r.stdout
    m = json.loads(subprocess.run([binary] + core + ["-nsam", "2", "-tmax", "4", "-dumpmodel"], capture_output=True, text=True).stdout)
    E = len(m["change_times"])
    model = dict(change_times=np.array(m["change_times"]), pop_sizes=np.array(m["pop_sizes"])[:, 0], lags=np.full(E, 20000.0),
                 nsam=2, loci_length=float(L), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"])
    S = segmod.Segments(seg, 2, L, max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    g = ParticleFilter(model, 300, seed=4, max_trace_events=0)
    g.init_prior(segs["start"][0]); g.load_segments(segs)
    g.load_lookahead(segmod.pack_lookahead(S.rows, 2), 2, pf.terminal_branch_quantiles(model, seed=1, n_trees=1000000))
    g.run(); g.finish()
    assert outfile.outfile_text(model, g.counts(), 300) == open(tmp_path / "apf.out").read()


# ---------------------------------------------------------------- local recombination map (.recomb.gz)

@pytest.mark.parametrize("n,E,Np,force_lds", [(4, 8, 600, False), (2, 4, 300, False), (6, 8, 256, True)])
def test_local_recombination_map_parity(oracle, hiplib, n, E, Np, force_lds):
    """record_local_recomb_events (count.cpp:559-613): the 100-bp differential opportunity and the per-sample / time /
    log-time counts equal the oracle's (sums of the same terms in another order: relative 1e-9 of the column scale)."""
    from smcsmc_amd import ParticleFilter
    model = cases.make_model(n=n, E=E, L=1.0e5)
    segs = cases.make_segments(model, seed=40 + n, max_seg_len=5000)
    o = oracle.Oracle(model, Np, seed=3); o.enable_local_recomb(); o.init_prior(0.0)
    o.run(o.pack_segments(model, segs))
    g = ParticleFilter(model, Np, seed=3, local_recomb=True, debug=1 if force_lds else 0)
    g.init_prior(0.0); g.load_segments(segs); g.run(); g.finish()
    assert _bits([o.logl()])[0] == _bits([g.logl()])[0]
    lo, lg = o.local_recomb(model["loci_length"]), g.local_recomb()
    # compare the absolute (cumulated) opportunity: the differential form cancels large terms
    co, cg = np.cumsum(lo["opp_diff"]), np.cumsum(lg["opp_diff"])
    assert co.max() > 0
    np.testing.assert_allclose(cg, co, rtol=1e-7, atol=1e-7 * co.max())
    assert lo["counts"][:n].sum() > 0
    np.testing.assert_allclose(lg["counts"], lo["counts"], rtol=1e-9, atol=1e-12 * max(1.0, lo["counts"].max()))
    # what the map means: per-interval opportunity = posterior mean tree length x 100 bp (x recorded epochs), and the
    # per-sample counts of an event sum to its posterior weight
    tot_rec = g.counts()["rec_count"].sum()
    assert lg["counts"][:n].sum() == pytest.approx(tot_rec, rel=1e-9)
    assert cg[:-1].sum() == pytest.approx(g.counts()["rec_opp"].sum(), rel=5e-3)      # up to the edge intervals


def test_binary_writes_the_local_recombination_map(hiplib, tmp_path):
    """<prefix>.recomb.gz (count.cpp:616-654) from the binary equals the table written from the same run through the
    python binding, and has the reference's header and one row per 100 bp."""
    import gzip
    import json
    import os
    import subprocess
    from smcsmc_amd import ParticleFilter, outfile, segments as segmod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    seg = os.path.join(root, "tests", "golden", "seg", "constpopsize_first3000.seg")
    L = 500000
    core = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.05 1 -eN 0.5 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
    r = subprocess.run([binary] + core + ["-nsam", "2", "-Np", "200", "-EM", "0", "-tmax", "4", "-lag", "10000", "-seed", "2",
                                          "-seg", seg, "-o", str(tmp_path / "m")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = gzip.open(tmp_path / "m.recomb.gz", "rt").read()
    lines = text.splitlines()
    assert lines[0].split("\t") == ["iter", "locus", "size", "opp_per_nt", "1", "2", "time", "log_time"]
    assert len(lines) == 1 + L // 100 and lines[1].split("\t")[1:3] == ["1", "100"]
    m = json.loads(subprocess.run([binary] + core + ["-nsam", "2", "-tmax", "4", "-dumpmodel"], capture_output=True, text=True).stdout)
    E = len(m["change_times"])
    model = dict(change_times=np.array(m["change_times"]), pop_sizes=np.array(m["pop_sizes"])[:, 0], lags=np.full(E, 10000.0),
                 nsam=2, loci_length=float(L), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"])
    S = segmod.Segments(seg, 2, L, max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    g = ParticleFilter(model, 200, seed=2, max_trace_events=0, local_recomb=True)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    mine = outfile.recomb_text(g.local_recomb(), 2).splitlines()
    # atomics make the last digits of a sum order-dependent: compare numerically at the printed precision
    a = np.array([[float(v) for v in ln.split("\t")] for ln in lines[1:]])
    b = np.array([[float(v) for v in ln.split("\t")] for ln in mine[1:]])
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-12)
    assert a[:, 3].max() > 0 and a[:, 4:6].sum() > 0


# ---------------------------------------------------------------- variational-Bayes event-count correction (-vb)

@pytest.mark.parametrize("n,P", [(4, 1), (10, 1), (4, 2)])
def test_variational_bayes_weight_factors_parity(oracle, hiplib, n, P):
    """model().variational_bayes_correction_ (particle.cpp:266-272): every coalescence / migration event multiplies the
    particle's weights by exp_digamma(c)/c for the event count c of its epoch (and populations)."""
    E = 6
    base = cases.make_model(n=n, E=E, L=8e4)
    segs = cases.make_segments(base, seed=60 + n, max_seg_len=5000)
    rng = np.random.default_rng(5)
    model = base if P == 1 else cases.make_structured(base, P=P, split_epoch=4, mig=2.0)
    model = dict(model, vb_coal_counts=rng.uniform(0.5, 40.0, (E, P)), vb_mig_counts=rng.uniform(0.5, 40.0, (E, P, P)))
    o, si, g = _run_both(oracle, model, segs, 300, seed=4)
    po, pg = o.particles(), g.particles()
    assert (_bits(po["w_post"]) == _bits(pg["w_post"])).all()            # the prior trees already carry factors
    assert np.ptp(po["w_post"]) > 0
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    # the correction changes the weights: the same run without it has another likelihood
    o0, si0, g0 = _run_both(oracle, {k: v for k, v in model.items() if not k.startswith("vb_")}, segs, 300, seed=4)
    g0.run(); g0.finish()
    assert g0.logl() != g.logl()


def test_device_exp_digamma(oracle, hiplib):
    """exp_digamma (particle.cpp:65-74) through the factor table the device builds: close to exp(psi(c))/c."""
    from scipy.special import digamma
    from smcsmc_amd import ParticleFilter
    c = np.array([[0.3], [1.0], [5.9], [6.0], [10.0], [10.5], [1e10]])
    model = dict(cases.make_model(n=2, E=7, L=1e4), vb_coal_counts=c)
    g = ParticleFilter(model, 64, seed=1); g.init_prior(0.0)
    w = g.particles()["w_post"] * 64                                     # one coalescence per prior tree: w = factor of its epoch
    expect = np.exp(digamma(c[:, 0])) / c[:, 0]
    for v in np.unique(w):
        assert np.min(np.abs(expect / v - 1)) < 2e-3                     # the reference's asymptotic series, not exact digamma


# ---------------------------------------------------------------- recombination guide (-guide)

def _guide(model, K, spread, seed):
    rng = np.random.default_rng(seed)
    L, n = model["loci_length"], model["nsam"]
    pos = np.floor(np.arange(K) * L / K)
    rates = model["recombination_rate"] * rng.uniform(1.0 / spread, spread, K)
    leaf = rng.uniform(1.0 / spread, spread, (K, n))
    leaf /= leaf.sum(1, keepdims=True)
    return dict(positions=pos, rates=rates, leaf_rates=leaf)


@pytest.mark.parametrize("n,bias", [(4, False), (4, True), (7, True), (2, False), (10, True), (12, False)])
def test_recombination_guide_parity(oracle, hiplib, n, bias):
    """Position-dependent sampling rate with per-sample relative rates (RecombinationBias, pfparam.hpp:96-223;
    samplePoint / importance_weight_over_segment / sampleNextBase, particle.cpp:942-1254), with and without the height
    bias: trees, weights, delayed factors, ESS and resampling indices bit-identical to the oracle."""
    E = 6
    model = cases.make_model(n=n, E=E, L=1.2e5)
    segs = cases.make_segments(model, seed=70 + n, max_seg_len=5000)
    extra = dict(guide=_guide(model, 9, 2.5, n), application_delays=np.full(E, 3000.0))
    if bias:
        extra.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0])
    model = dict(model, **extra)
    o, si, g = _run_both(oracle, model, segs, 400 if n <= 8 else 160, seed=6)      # n > 8: the LDS-tree kernel
    o.run(si); g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert (to["resampled"] == tg["resampled"]).all() and to["resampled"].sum() > 0
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    so, po_ = o.resample_events(); sg_, pg_ = g.resample_events()
    assert (so == sg_).all() and (po_ == pg_).all()
    _assert_state_equal(o, g)
    _assert_counts_close(o.counts(), g.counts())


def test_recombination_guide_is_an_importance_sampler_of_the_model(hiplib):
    """Without data the weights of the guided sampler average to one and the lagged counts recover the model's
    rates -- the true recombination rate, not the guide's (importance_weight_over_segment + the event weights)."""
    from smcsmc_amd import ParticleFilter
    E = 6
    model = cases.make_model(n=4, E=E, L=2e6)
    model = dict(model, guide=_guide(model, 8, 1.4, 1), application_delays=np.full(E, 5000.0))
    assert abs(model["guide"]["rates"].mean() / 1e-8 - 1) > 0.02          # the guide is off on average
    segs = cases.nodata_segments(model, 4000.0)
    g = ParticleFilter(model, 4096, seed=3); g.init_prior(0.0); g.load_segments(segs); g.run(); g.finish()
    c = g.counts()
    assert abs(c["rec_count"].sum() / c["rec_opp"].sum() / 1e-8 - 1) < 0.01
    coal = c["coal_count"][2:5] / c["coal_opp"][2:5] * 2e4
    assert np.abs(coal - 1).max() < 0.03
    assert abs(g.logl()) < 0.5


def test_binary_recombination_guide(hiplib, tmp_path):
    """bin/smcsmc -guide file equals the library run with the guide read by the Python mirror and the application
    delays from the calibration at the true rate (smcsmc.cpp:287-307), character for character."""
    import json
    import os
    import subprocess
    from smcsmc_amd import ParticleFilter, outfile, pf, segments as segmod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    seg = os.path.join(root, "tests", "golden", "seg", "constpopsize_first3000.seg")
    L = 1000000
    rng = np.random.default_rng(5)
    guide = tmp_path / "guide.txt"
    with open(guide, "w") as f:
        f.write("locus\tsize\trecomb_rate\t1\t2\n")
        for k in range(10):
            lr = rng.uniform(0.5, 2.0, 2)
            f.write("%d\t%d\t%.6g\t%s\n" % (k * 100000, 100000, 1e-8 * rng.uniform(0.6, 1.6), "\t".join("%.4f" % v for v in lr / lr.sum())))
    core = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.01 1 -eN 0.25 1 -eN 1 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
    r = subprocess.run([binary] + core + ["-nsam", "2", "-Np", "300", "-EM", "0", "-tmax", "4", "-lag", "20000", "-seed", "4",
                                          "-guide", str(guide), "-seg", seg, "-o", str(tmp_path / "gd")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Setting model rates" in r.stdout
    m = json.loads(subprocess.run([binary] + core + ["-nsam", "2", "-tmax", "4", "-dumpmodel"], capture_output=True, text=True).stdout)
    E = len(m["change_times"])
    model = dict(change_times=np.array(m["change_times"]), pop_sizes=np.array(m["pop_sizes"])[:, 0], lags=np.full(E, 20000.0),
                 nsam=2, loci_length=float(L), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"])
    med, _ = pf.median_survival(model, seed=1, min_events=200, max_trees=1000000)
    model = dict(model, guide=segmod.read_guide(str(guide), 2), application_delays=med * 0.5)
    S = segmod.Segments(seg, 2, L, max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    g = ParticleFilter(model, 300, seed=4, max_trace_events=0)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    assert outfile.outfile_text(model, g.counts(), 300) == open(tmp_path / "gd.out").read()


# ---------------------------------------------------------------- -arg: tree dump of one sampled particle

def _replay_tree_events(n, kind, pos, height, desc):
    """Rebuilds the local tree from the events of a tree dump, first position first (what smcsmc/trees2tskit.py:125-190
    does with them): every node is (height, samples below).  Returns the node list of the last tree."""
    order = list(range(len(kind)))[::-1]               # the dump runs from the last position back to the first
    nodes = []                                         # (height, mask), the coalescent nodes of the current tree
    i = 0
    while i < len(order):
        c = order[i]                                   # read backwards, a coalescence comes before its recombination
        assert kind[c] == 1
        if i + 1 >= len(order) or kind[order[i + 1]] != 0:
            nodes.append((height[c], int(desc[c])))    # initial tree: a new sample joins, no recombination
            i += 1
            continue
        e = order[i + 1]
        cut, h = int(desc[e]), height[e]               # R: remove the cut samples from every node above the cut ...
        assert pos[c] == pos[e] and height[c] > h
        new_mask, tc = int(desc[c]), height[c]
        assert new_mask & cut == cut
        pruned = []
        for (t, m) in nodes:
            if t > h and (m & cut) == cut:
                m &= ~cut
            pruned.append((t, m))
        # ... a node left with a single subtree below it disappears (the old parent of the cut branch)
        keep = []
        for (t, m) in pruned:
            below = [mm for (tt, mm) in pruned if tt < t and (mm & m) == mm and mm]
            covered = 0
            for mm in below:
                covered |= mm
            leaves = m & ~covered
            nsub = bin(leaves).count("1") + len([mm for mm in below if not any((mm & m2) == mm and m2 != mm and (m2 & m) == m2 for (t2, m2) in pruned if t2 < t)])
            keep.append((t, m, nsub))
        pruned = [(t, m) for (t, m, nsub) in keep if nsub >= 2 and m]
        # ... and the new node adds them to everything above it on the path to the root
        target = new_mask & ~cut
        if target == 0:                                # back into its own branch: the old tree returns
            nodes = sorted(nodes)
            i += 2
            continue
        out = []
        for (t, m) in pruned:
            if t > tc and (m & target) == target:
                m |= cut
            out.append((t, m))
        out.append((tc, new_mask))
        nodes = sorted(out)
        i += 2
    return sorted(nodes)


@pytest.mark.parametrize("n,Np,force_lds,bias", [(4, 300, False, False), (6, 200, False, False), (5, 150, True, False), (2, 100, False, False),
                                                 (8, 200, False, True), (7, 120, True, True)])
def test_tree_dump_of_the_sampled_particle(oracle, hiplib, n, Np, force_lds, bias):
    """-arg (pc.cpp:515-555): replaying the dumped events of the drawn particle's history from the first position on must
    end in that particle's own local tree (node heights and the samples below each node); every recombination is
    followed by its coalescence at the same position, above the cut; the descendants of a coalescence contain those of
    its recombination."""
    from smcsmc_amd import ParticleFilter, outfile
    model = cases.make_model(n=n, E=8, L=1.5e5)
    segs = cases.make_segments(model, seed=80 + n, max_seg_len=5000)
    if bias:
        model = dict(model, bias_heights=[400.0], bias_strengths=[4.0, 1.0], application_delays=np.full(8, 3000.0))
    g = ParticleFilter(model, Np, seed=4, max_trace_events=0, record_trees=True, log_cap=8192, gen_cap=4096,
                       debug=1 if force_lds else 0)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    assert g.trace()["resampled"].sum() > 3
    part, kind, pos, hgt, desc = g.sample_tree_events()
    assert 0 <= part < Np and len(kind) > 2 * (n - 1)
    # the oracle keeps the reference's own structure, a linked list of tree events per particle shared with its copies
    o = oracle.Oracle(model, Np, seed=4, max_trace_events=0)
    o.enable_tree_recording()
    o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
    opart, okind, opos, ohgt, odesc = o.sample_tree_events()
    assert opart == part and (okind == kind).all() and (odesc == desc).all()
    assert (_bits(opos) == _bits(pos)).all() and (_bits(ohgt) == _bits(hgt)).all()
    assert (np.diff(pos) <= 0).all()                                        # last position first
    nodes = _replay_tree_events(n, kind, pos, hgt, desc)
    p = g.particles()
    S = p["heights"][part]; Cc = p["children"][part].reshape(n - 1, 2)
    masks = []
    for r in range(n - 1):
        m = 0
        for c in Cc[r]:
            m |= (1 << int(c)) if c < n else masks[int(c) - n]
        masks.append(m)
    expect = sorted((float(S[r]), masks[r]) for r in range(n - 1))
    assert [m for _, m in nodes] == [m for _, m in expect]
    np.testing.assert_allclose([t for t, _ in nodes], [t for t, _ in expect], rtol=0, atol=0)
    # the text form: one decimal, positions shifted by the start position
    text = outfile.trees_text(kind, pos, hgt, desc, start_position=1.0)
    first = text.splitlines()[0].split("\t")
    assert first[0] in "RC" and float(first[1]) == pytest.approx(pos[0], abs=0.05) and len(text.splitlines()) == len(kind)
    assert outfile.descendants_text(0b0101) == "101" and outfile.descendants_text(0) == "0" and outfile.descendants_text(0b1000) == "0001"


def test_binary_writes_the_tree_dump(hiplib, tmp_path):
    """bin/smcsmc -arg writes <prefix>.trees.gz: the lines the python mirror makes from the same run, character for
    character, in the reference's format (code, position, height, from, to, descendants; pc.cpp:515-555)."""
    import gzip
    import json
    import os
    import subprocess
    from smcsmc_amd import ParticleFilter, outfile, segments as segmod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    seg = os.path.join(root, "tests", "golden", "seg", "constpopsize_first3000.seg")
    L = 1000000
    core = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.01 1 -eN 0.25 1 -eN 1 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
    r = subprocess.run([binary] + core + ["-nsam", "2", "-Np", "200", "-EM", "0", "-tmax", "4", "-lag", "20000", "-seed", "4",
                                          "-arg", "-seg", seg, "-o", str(tmp_path / "arg")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = gzip.open(tmp_path / "arg.trees.gz", "rt").read()
    lines = text.splitlines()
    assert len(lines) > 10 and all(ln.split("\t")[0] in ("R", "C") and len(ln.split("\t")) == 6 for ln in lines)
    import trees_format
    trees_format.check_lines(lines, nsam=2, npop=1)           # the grammar of the reference's own example file
    assert lines[-1].split("\t")[0] == "C" and lines[-1].split("\t")[5] == "11"       # the first tree: sample 2 joins sample 1
    m = json.loads(subprocess.run([binary] + core + ["-nsam", "2", "-tmax", "4", "-dumpmodel"], capture_output=True, text=True).stdout)
    E = len(m["change_times"])
    model = dict(change_times=np.array(m["change_times"]), pop_sizes=np.array(m["pop_sizes"])[:, 0], lags=np.full(E, 20000.0),
                 nsam=2, loci_length=float(L), mutation_rate=m["mutation_rate"], recombination_rate=m["recombination_rate"])
    S = segmod.Segments(seg, 2, L, max_segment_length=int(2.0 / (m["recombination_rate"] * 4 * m["N0"])))
    segs = S.pack(model["lags"])
    g = ParticleFilter(model, 200, seed=4, max_trace_events=0, local_recomb=True, record_trees=True, log_cap=16384, gen_cap=8192)
    g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
    _, kind, pos, hgt, desc = g.sample_tree_events()
    assert outfile.trees_text(kind, pos, hgt, desc, start_position=1.0) == text


def test_binary_flag_combinations(hiplib, tmp_path):
    """The flag combinations a front-end or a user can ask for run through, and none of the sampling aids moves the
    likelihood estimate of the same data by more than a per cent."""
    import os
    import subprocess
    from smcsmc_amd import outfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "bin", "smcsmc")
    seg = os.path.join(root, "tests", "golden", "seg", "constpopsize_first3000.seg")
    L = 1000000
    core = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.01 1 -eN 0.25 1 -eN 1 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
    common = ["-nsam", "2", "-seg", seg, "-Np", "300", "-tmax", "4", "-seed", "3"]
    env = dict(os.environ)
    runs = {
        "plain": [],
        "apf_bias": ["-apf", "2", "-bias_heights", "400", "-bias_strengths", "3", "1"],
        "arg_apf": ["-apf", "1", "-arg"],
        "arg_bias_xr": ["-arg", "-bias_heights", "400", "-bias_strengths", "3", "1", "-xr", "0-1", "-delay_coal"],
        "em_arg": ["-arg", "-EM", "1"],
        "ess_record": ["-ESS", "0.7", "-record_ess", "-calibrate_lag", "1.5"],
    }
    ll = {}
    for name, extra in runs.items():
        r = subprocess.run([binary] + core + common + extra + ["-o", str(tmp_path / name)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (name, r.stderr[-300:])
        rows = [ln.split() for ln in open(tmp_path / (name + ".out")).read().splitlines()[1:]]
        ll[name] = float([x for x in rows if x[4] == "LogL" and x[0] == "0"][0][8])
        if "-arg" in extra:
            assert os.path.getsize(tmp_path / (name + ".trees.gz")) > 100
    assert os.path.exists(tmp_path / "ess_record.resample")
    for name, v in ll.items():
        assert abs(v / ll["plain"] - 1) < 0.01, (name, v, ll["plain"])
