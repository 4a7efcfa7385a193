"""GPU parity at the sizes BASELINE.json quotes its metric on, and with the device rings wrapping.

The sweeps of test_gpu_parity.py stop at Np = 1000; the headline configurations run 10 000 (C3) and 20 000 (C5)
particles, where the launch has 40 / 313 workgroups, the canonical reduction has three levels, the last wavefront is
partial and the newest run list of the ancestor ledger (about 0.55 Np survivors) fills a whole 1024-thread tile.  The
oracle manages about 20 rows per second at that size, so these tests compare the first few hundred rows of a chunk of
the bench's own shape -- rows, weights, ESS, log-likelihood, resampling indices, the particle states at the end and
the lagged counts.

The event log (records per particle slot) and the ancestor ledger (generations) are rings.  A 100 Mb sweep wraps both
several times; the tests below make them wrap more than ten times on a sweep the oracle finishes in seconds.
"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

COUNT_RTOL = 1e-9


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _bench_model(n, E, L, pops=1):
    """The model bench.py builds (N0 1e4, mu 2.5e-8, rho 1e-8, E log-spaced epochs, the reference's uncalibrated lags)."""
    from smcsmc_amd import simulate
    N0, rho = 1e4, 1e-8
    ct = simulate.default_epochs(E)
    lags = np.array([4.0 / (rho * (ct[e + 1] if e + 1 < E else ct[-1])) for e in range(E)])
    model = dict(change_times=ct, pop_sizes=np.full(E, N0), lags=lags, nsam=n, loci_length=float(L),
                 mutation_rate=2.5e-8, recombination_rate=rho)
    if pops > 1:
        P = pops
        split = int(np.searchsorted(ct, 0.5 * 4 * N0))
        split = min(max(split, 1), E - 1)
        mr = np.zeros((E, P, P)); sm = np.zeros((E, P, P))
        for e in range(split):
            mr[e] = (1.0 / (4 * N0)) * (1 - np.eye(P))
        sm[split, 1:, 0] = 1.0
        model.update(n_pops=P, pop_sizes=np.repeat(model["pop_sizes"][:, None], P, axis=1), mig_rates=mr, single_mig=sm,
                     sample_pops=[i * P // n for i in range(n)])
    return model


def _compare_sweep(oracle, model, segs, Np, seed, structured=False, **gpu_kw):
    from smcsmc_amd import ParticleFilter
    o = oracle.Oracle(model, Np, seed=seed, max_trace_events=64)
    o.init_prior(segs["start"][0])
    si = o.pack_segments(model, segs)
    g = ParticleFilter(model, Np, seed=seed, max_trace_events=64, **gpu_kw)
    g.init_prior(segs["start"][0]); g.load_segments(segs)
    o.run(si)
    g.run(); g.finish()
    to, tg = o.trace(), g.trace()
    assert g.segments_done() == len(to["T"])
    assert (to["resampled"] == tg["resampled"]).all()
    for k in ("T", "ess", "logl"):
        assert (_bits(to[k]) == _bits(tg[k])).all(), k
    so, po = o.resample_events(); sg, pg = g.resample_events()
    assert (so == sg).all() and (po == pg).all()                 # resampling indices bit-exact
    ps_o, ps_g = o.particles(), g.particles()
    assert (ps_o["children"] == ps_g["children"]).all()
    for k in ("heights", "w_post", "w_pilot", "next_base"):
        assert (_bits(ps_o[k]) == _bits(ps_g[k])).all(), k
    assert _bits([o.logl()])[0] == _bits([g.logl()])[0]
    co, cg = o.counts(), g.counts()
    keys = ["coal_count", "coal_opp", "coal_weight", "rec_count", "rec_opp", "rec_weight"]
    if structured:
        keys += ["mig_count", "mig_opp", "mig_weight"]
    for k in keys:
        np.testing.assert_allclose(cg[k], co[k], rtol=COUNT_RTOL, atol=1e-300, err_msg=k)
    assert cg["resample_count"] == co["resample_count"]
    return to, co, g


def test_c3_shape_at_its_own_particle_count(oracle, hiplib):
    """BASELINE.json configs[2]: 2 diploids, Np = 10 000, E = 32 -- the first 160 kb (about 300 rows) of a chunk."""
    model = _bench_model(4, 32, 1.6e5)
    segs = cases.make_segments(model, seed=1, max_seg_len=5000)
    assert len(segs["start"]) >= 250
    to, co, g = _compare_sweep(oracle, model, segs, 10000, seed=1)
    assert to["resampled"].sum() >= 50, "the ESS test must fire often at this size (42 % of the rows on the full chunk)"
    assert co["rec_count"].sum() > 0


def test_c3_on_the_configuration_bench_py_times(oracle, hiplib):
    """The same check on exactly what bench.py runs by default: its build_workload() -- data drawn on the device (k_simulate),
    the calibrated lags of the binary's default (-calibrate_lag 2) -- at Np = 10 000; a 160 kb chunk of it."""
    import argparse
    import bench
    args = argparse.Namespace(nsam=4, length=1.6e5, epochs=32, pops=1, np=10000, device=0, host_data=False, uncalibrated_lags=False)
    model, segs = bench.build_workload(args, seed=1)
    uncal = _bench_model(4, 32, 1.6e5)["lags"]
    assert not np.allclose(model["lags"], uncal)               # calibrated, not the round-1 defaults
    assert len(segs["start"]) >= 200
    to, co, g = _compare_sweep(oracle, model, segs, 10000, seed=1, local_recomb=True)
    assert to["resampled"].sum() >= 40
    assert co["rec_count"].sum() > 0


def test_c5_shape_at_its_own_particle_count(oracle, hiplib):
    """BASELINE.json configs[4]: 4 diploids from two populations (isolation with migration), Np = 20 000, E = 32 -- the
    first 40 kb (about 100 rows)."""
    model = _bench_model(8, 32, 4.0e4, pops=2)
    segs = cases.make_segments(dict(model, pop_sizes=model["pop_sizes"][:, 0]), seed=2, max_seg_len=5000)
    assert len(segs["start"]) >= 80
    to, co, g = _compare_sweep(oracle, model, segs, 20000, seed=2, structured=True)
    assert to["resampled"].sum() >= 10
    assert co["mig_count"].sum() > 0


@pytest.mark.parametrize("n,E,Np,P,debug", [(4, 8, 500, 1, 0), (4, 8, 500, 1, 8), (4, 8, 500, 1, 2), (6, 8, 320, 1, 1), (4, 6, 300, 2, 0)])
def test_rings_wrap_many_times(oracle, hiplib, n, E, Np, P, debug):
    """Event log of 64 records per slot, ledger of 32 generations (the defaults are 16 384 and 8 192; a C3 sweep wraps
    them about 6 and 10 times): both wrap more than ten times here and nothing changes -- every compared quantity is
    still the oracle's, which has no rings at all (ref-counted event chains, as in the reference)."""
    base = cases.make_model(n=n, E=E, L=6.0e5, lag=2500.0)
    segs = cases.make_segments(base, seed=90 + n + P, max_seg_len=5000)
    model = base if P == 1 else cases.make_structured(base, P=P, split_epoch=E - 3, mig=2.0)
    kw = dict(log_cap=64, gen_cap=32, debug=debug)
    if P > 1:
        kw["piece_cap"] = 256
    to, co, g = _compare_sweep(oracle, model, segs, Np, seed=5, structured=P > 1, **kw)
    st = g.stats()
    nres = int(to["resampled"].sum())
    assert nres > 10 * 32, "ledger ring must wrap more than ten times (%d generations)" % nres
    assert st["records"] / Np > 10 * 64 * (0.5 if P > 1 else 1.0), "event log must wrap many times (%.0f records per slot)" % (st["records"] / Np)


@pytest.mark.parametrize("P,log_cap,gen_cap,what", [(1, 8, 4096, "event log ring overflow"), (1, 4096, 4, "generation ledger overflow"),
                                                    (2, 16, 4096, "event log ring overflow"), (2, 4096, 24, "generation ledger overflow"),
                                                    (2, 4096, 8, "gen_cap must be at least 20")])
def test_ring_too_small_is_a_reported_error(hiplib, P, log_cap, gen_cap, what):
    """A ring that cannot hold what the lags keep alive must stop the run with an error, never overwrite silently -- also where
    the extend launches run up to fourteen rows ahead of the bookkeeping and the counts (structured models on the row pipeline):
    there the generation ledger keeps sixteen entries of headroom and the extend role checks its own writes to the event log."""
    from smcsmc_amd import ParticleFilter
    from smcsmc_amd.pf import PfError
    model = cases.make_model(n=4, E=8, L=3.0e5, lag=60000.0)
    segs = cases.make_segments(model, seed=12, max_seg_len=5000)
    if P > 1:
        model = cases.make_structured(model, P=P, split_epoch=5, mig=1.0)
    with pytest.raises(PfError, match=what):
        g = ParticleFilter(model, 400, seed=3, log_cap=log_cap, gen_cap=gen_cap)
        g.init_prior(0.0); g.load_segments(segs)
        g.run(); g.finish()
