"""The device-side data simulator (k_simulate / pf_simulate_sites): synthetic `.seg` data from the same SMC' process the
filter simulates (the reference shells out to scrm, populationmodels.py:440-577)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def test_device_simulator_matches_coalescent_expectations(hiplib):
    from smcsmc_amd import simulate
    n, L, N0, mu, rho = 4, 2.0e6, 1e4, 2.5e-8, 1e-8
    ct = simulate.default_epochs(8)
    chunks = simulate.simulate_seg_device(n, L, mu, rho, ct, np.full(8, N0), seed=11, nchunks=24)
    assert len(chunks) == 24
    # segregating sites: E[S] = 4 N mu L H(n-1); the chunks are independent replicates
    S = np.array([len(c["start"]) - 1 for c in chunks], float)
    expect = 4 * N0 * mu * L * (1 + 0.5 + 1.0 / 3)
    assert abs(S.mean() / expect - 1) < 0.04, (S.mean(), expect)
    assert S.std() > 0.005 * expect and len({int(v) for v in S}) > 12          # replicates differ
    # site frequency spectrum: E[xi_i] proportional to 1/i
    al = np.concatenate([c["alleles"][:-1] for c in chunks])
    assert set(np.unique(al)) <= {0, 1}
    k = al.sum(axis=1)
    assert k.min() >= 1 and k.max() <= n - 1
    sfs = np.array([(k == i).sum() for i in (1, 2, 3)], float)
    np.testing.assert_allclose(sfs / sfs.sum(), np.array([1, 0.5, 1.0 / 3]) / (11.0 / 6), atol=0.012)
    # rows are well formed: consecutive, last row all missing and reaching the end
    c0 = chunks[0]
    assert (c0["start"][1:] == c0["start"][:-1] + c0["length"][:-1]).all() and c0["start"][0] == 1
    assert (c0["alleles"][-1] == -1).all() and c0["start"][-1] + c0["length"][-1] == int(L)
    # linkage: neighbouring sites share their tree more often than distant ones (recombination is simulated)
    same_near = (al[1:] == al[:-1]).all(axis=1).mean()
    same_far = (al[200:] == al[:-200]).all(axis=1).mean()
    assert same_near > same_far + 0.05


def test_device_simulator_is_reproducible_and_feeds_the_filter(hiplib, oracle):
    from smcsmc_amd import ParticleFilter, segments as segmod, simulate
    model = cases.make_model(n=4, E=8, L=3e5)
    a = simulate.simulate_seg_device(4, 3e5, 2.5e-8, 1e-8, model["change_times"], model["pop_sizes"], seed=5, nchunks=2)
    b = simulate.simulate_seg_device(4, 3e5, 2.5e-8, 1e-8, model["change_times"], model["pop_sizes"], seed=5, nchunks=2)
    for x, y in zip(a, b):
        assert all((x[k] == y[k]).all() for k in x)
    assert len(a[0]["start"]) != len(a[1]["start"]) or (a[0]["start"] != a[1]["start"]).any()
    segs = segmod.Segments.from_sites(a[0]["start"], a[0]["length"], a[0]["alleles"], 4, 3e5, max_segment_length=5000).pack(model["lags"])
    g = ParticleFilter(model, 500, seed=3); g.init_prior(0.0); g.load_segments(segs); g.run(); g.finish()
    o = oracle.Oracle(model, 500, seed=3); o.init_prior(0.0); o.run(o.pack_segments(model, segs))
    assert np.float64(g.logl()).view(np.uint64) == np.float64(o.logl()).view(np.uint64)
    assert np.isfinite(g.logl()) and g.logl() < 0


def test_structured_device_simulation_shows_the_split(hiplib):
    """k_simulate_mp: data from an isolation model (two populations, no migration, joined at T generations).  A pair
    of samples from one population coalesces after 2N generations on average, a pair across the populations after
    T + 2N: mean pairwise differences per base 4 N mu within and 2 mu (T + 2N) across."""
    from smcsmc_amd import simulate
    N0, mu, rho, L, n = 1e4, 2.5e-8, 1e-8, 4e6, 4
    ct = np.array([0.0, 5000.0, 20000.0, 60000.0])
    E, P = len(ct), 2
    T = ct[2]
    sm = np.zeros((E, P, P)); sm[2, 1, 0] = 1.0
    structure = dict(n_pops=P, pop_sizes=np.full((E, P), N0), mig_rates=np.zeros((E, P, P)), single_mig=sm, sample_pops=[0, 0, 1, 1])
    within, across = [], []
    for seg in simulate.simulate_seg_device(n, L, mu, rho, ct, np.full(E, N0), seed=3, nchunks=6, structure=structure):
        a = seg["alleles"][seg["alleles"].max(axis=1) >= 0]          # rows that carry a site
        a = a[(a.min(axis=1) == 0) & (a.max(axis=1) == 1)]
        d = lambda i, j: float((a[:, i] != a[:, j]).sum()) / L      # noqa: E731
        within += [d(0, 1), d(2, 3)]
        across += [d(0, 2), d(0, 3), d(1, 2), d(1, 3)]
    assert np.mean(within) == pytest.approx(4 * N0 * mu, rel=0.08)
    assert np.mean(across) == pytest.approx(2 * mu * (T + 2 * N0), rel=0.08)
    # with migration the populations look alike again
    structure["mig_rates"] = np.zeros((E, P, P))
    structure["mig_rates"][:2] = (8.0 / (4 * N0)) * (1 - np.eye(P))            # until the join
    seg = simulate.simulate_seg_device(n, L, mu, rho, ct, np.full(E, N0), seed=4, nchunks=1, structure=structure)[0]
    a = seg["alleles"]; a = a[(a.min(axis=1) == 0) & (a.max(axis=1) == 1)]
    w = np.mean([(a[:, 0] != a[:, 1]).mean(), (a[:, 2] != a[:, 3]).mean()]); x = np.mean([(a[:, 0] != a[:, 2]).mean(), (a[:, 1] != a[:, 3]).mean()])
    assert 0.9 < x / w < 1.3                      # against 2.0 without migration


def test_device_simulator_against_the_independent_numpy_simulator(hiplib):
    """k_simulate shares the filter's own transition code; smcsmc_amd.simulate.simulate_seg is an independent numpy
    implementation of the same SMC' process with mutations.  200 chunks of 50 kb from each under a model with a sixfold
    size change: two-sample Kolmogorov-Smirnov tests on the segregating sites per chunk, on the mean pairwise difference per
    chunk and on two linkage statistics (incompatible neighbouring sites: four-gamete test at lag 1 and at lag 10)."""
    from scipy import stats
    from smcsmc_amd import simulate
    n, L, N0, mu, rho = 4, 5.0e4, 1e4, 2.5e-8, 1e-8
    ct = np.array([0.0, 2000.0, 20000.0, 60000.0])
    ps = np.array([1.0, 0.3, 2.0, 1.0]) * N0
    R = 200
    dev = simulate.simulate_seg_device(n, L, mu, rho, ct, ps, seed=77, nchunks=R)
    host = [simulate.simulate_seg(n, L, mu, rho, ct, ps, seed=1000 + r) for r in range(R)]

    def four_gamete_failures(a, lag):
        if len(a) <= lag:
            return 0.0
        x, y = a[:-lag], a[lag:]
        bad = 0
        for i in range(len(x)):
            bad += len({(int(p), int(q)) for p, q in zip(x[i], y[i])}) == 4
        return bad / len(x)

    def stats_of(chunks):
        S, pi, g1, g10 = [], [], [], []
        for c in chunks:
            a = c["alleles"][:-1]
            a = a[(a.min(axis=1) >= 0)]
            S.append(len(a))
            d = 0.0
            for i in range(n):
                for j in range(i + 1, n):
                    d += (a[:, i] != a[:, j]).sum()
            pi.append(d / (n * (n - 1) / 2) / L)
            g1.append(four_gamete_failures(a, 1)); g10.append(four_gamete_failures(a, 10))
        return [np.array(v, float) for v in (S, pi, g1, g10)]

    sd, sh = stats_of(dev), stats_of(host)
    for name, x, y in zip(("segregating sites", "pairwise difference", "four-gamete lag 1", "four-gamete lag 10"), sd, sh):
        p = stats.ks_2samp(x, y).pvalue
        assert p > 1e-3, (name, p, x.mean(), y.mean())
    # and the two agree on the means within their standard errors (3.5 sigma)
    for x, y in zip(sd[:2], sh[:2]):
        se = np.sqrt(x.var() / R + y.var() / R)
        assert abs(x.mean() - y.mean()) < 3.5 * se, (x.mean(), y.mean(), se)
