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


def _run_both(oracle, model, segs, Np, seed, ess=0.5):
    from smcsmc_amd import ParticleFilter
    o = oracle.Oracle(model, Np, ess_fraction=ess, seed=seed, max_trace_events=64)
    o.init_prior(segs["start"][0])
    si = o.pack_segments(model, segs)
    g = ParticleFilter(model, Np, ess_fraction=ess, seed=seed, max_trace_events=64)
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


@pytest.mark.parametrize("n,E,P,Np,seed", [(4, 8, 2, 600, 2), (8, 8, 2, 256, 3), (6, 6, 3, 300, 4)])
def test_structured_full_sweep_parity(oracle, hiplib, n, E, P, Np, seed):
    base = cases.make_model(n=n, E=E, L=1.0e5)
    segs = cases.make_segments(base, seed=seed, max_seg_len=5000)
    model = cases.make_structured(base, P=P, split_epoch=E - 3, mig=2.0)
    o, si, g = _run_both(oracle, model, segs, Np, seed)
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
    coal = c["coal_count"] / c["coal_opp"] * 2 * N0
    mig = c["mig_count"].sum(2) / c["mig_opp"] * 4 * N0
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
