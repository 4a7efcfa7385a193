"""Randomised parity sweep: model shape (haplotypes, epochs with random sizes, populations with migration), particle
count, ESS threshold, phasing and sampling mode (plain, focused sampling, recombination guide, both, variational-Bayes
factors, auxiliary particle filter) drawn at random; every run must match the oracle bit for bit in trees, weights, ESS,
log-likelihood and resampling flags, and to 1e-8 of the column scale in the lagged counts."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_configurations_match_the_oracle(oracle, hiplib, seed):
    from smcsmc_amd import ParticleFilter, pf as pfm, segments as segmod
    rng = np.random.default_rng(seed)
    done = 0
    for it in range(14):
        n = int(rng.integers(2, 13)); E = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 63, 64])); P = int(rng.choice([1, 1, 1, 2, 3]))
        if P > 1:
            n, E = max(n, P), max(E, 4)
        L = float(rng.choice([4e4, 8e4, 1.5e5])); Np = int(rng.choice([1, 33, 64, 100, 257, 400]))
        base = cases.make_model(n=n, E=E, L=L, sizes=rng.uniform(0.3, 3.0, E))
        model = cases.make_structured(base, P=P, mig=float(rng.choice([0.2, 1.0, 2.0]))) if P > 1 else base
        mode = str(rng.choice(["plain", "bias", "guide", "guide+bias", "vb", "apf", "apf"]))
        extra = {}
        if mode in ("bias", "guide+bias"):
            extra.update(bias_heights=[float(rng.choice([200.0, 400.0, 3000.0]))],
                         bias_strengths=[float(rng.choice([2.0, 5.0, 10.0])), 1.0], delay_type=int(rng.integers(0, 3)))
        if mode in ("guide", "guide+bias"):
            K = int(rng.integers(1, 7)); leaf = rng.uniform(0.3, 3.0, (K, n)); leaf /= leaf.sum(1, keepdims=True)
            extra.update(guide=dict(positions=np.floor(np.arange(K) * L / K), rates=1e-8 * rng.uniform(0.3, 3.0, K), leaf_rates=leaf))
        if mode in ("bias", "guide", "guide+bias"):
            extra.update(application_delays=np.full(E, float(rng.choice([0.5, 2000.0, 8000.0]))))
        if mode == "vb":
            extra.update(vb_coal_counts=rng.uniform(0.5, 50.0, (E, P)), vb_mig_counts=rng.uniform(0.5, 50.0, (E, P, P)))
        model = dict(model, **extra)
        segs = cases.make_segments(base, seed=int(rng.integers(1, 10**6)), unphased=bool(rng.integers(0, 2)),
                                   max_seg_len=int(rng.choice([2000, 5000])))
        run_seed = int(rng.integers(1, 10**6)); essf = float(rng.choice([0.0, 0.5, 0.9]))
        tag = "seed %d it %d: n %d E %d P %d Np %d L %g %s ess %.1f" % (seed, it, n, E, P, Np, L, mode, essf)
        try:
            o = oracle.Oracle(model, Np, ess_fraction=essf, seed=run_seed, max_trace_events=32)
        except RuntimeError:
            continue
        try:
            o.init_prior(segs["start"][0]); si = o.pack_segments(model, segs)
            g = ParticleFilter(model, Np, ess_fraction=essf, seed=run_seed, max_trace_events=32)
            g.init_prior(segs["start"][0]); g.load_segments(segs)
            if mode == "apf":
                rows = [(int(s_) + 1, int(l_), int(st_), list(map(int, a_)))
                        for s_, l_, st_, a_ in zip(segs["start"], segs["length"], segs["state"], segs["alleles"])]
                la = segmod.pack_lookahead(rows, n); tbl = pfm.terminal_branch_quantiles(model, seed=1, n_trees=8000)
                lvl = int(rng.integers(1, 5))
                o.load_lookahead(la, lvl, tbl); g.load_lookahead(la, lvl, tbl)
            o.run(si)
        except RuntimeError as e:          # a capacity limit of the restatement (96 migration events per local tree)
            assert "too many migration events" in str(e), tag
            continue
        g.run(); g.finish()
        to, tg = o.trace(), g.trace()
        assert (to["resampled"] == tg["resampled"]).all(), tag
        for k in ("T", "ess", "logl"):
            assert (_bits(to[k]) == _bits(tg[k])).all(), (tag, k)
        po, pg = o.particles(), g.particles()
        assert (po["children"] == pg["children"]).all(), tag
        for k in ("heights", "w_post", "w_pilot", "next_base"):
            assert (_bits(po[k]) == _bits(pg[k])).all(), (tag, k)
        co, cg = o.counts(), g.counts()
        for k in ("coal_count", "coal_opp", "rec_count", "rec_opp"):
            np.testing.assert_allclose(cg[k], co[k], rtol=1e-8, atol=1e-8 * max(1e-300, np.abs(co[k]).max()), err_msg=tag + " " + k)
        g.close(); o.close()
        done += 1
    assert done >= 10


@pytest.mark.parametrize("seed", [0, 1])
def test_random_structured_configurations_match_the_oracle(oracle, hiplib, seed):
    """The structured row kernels (register tree, with the previous row completed while loading and as three launches)
    on random models: 2-4 populations with random sizes, symmetric or one-way migration at rates up to 10 (4 N0 m), a join
    at a random epoch or none, 4-64 epochs, 2-8 haplotypes, plain / focused sampling with every delay type /
    variational-Bayes factors.  Trees, node populations, migration event lists, weights, ESS, log-likelihood and
    resampling bit for bit; lagged counts including the migration statistics to 1e-8 of the column scale."""
    from smcsmc_amd import ParticleFilter
    rng = np.random.default_rng(1000 + seed)
    done = 0
    for it in range(8):
        P = int(rng.choice([2, 2, 3, 4])); n = int(rng.integers(max(2, P), 9)); E = int(rng.choice([4, 5, 8, 17, 32, 64]))
        L = float(rng.choice([4e4, 8e4, 1.5e5])); Np = int(rng.choice([33, 64, 100, 257, 400]))
        base = cases.make_model(n=n, E=E, L=L, sizes=rng.uniform(0.3, 3.0, E))
        split = int(rng.integers(2, E)) if rng.random() < 0.8 else E
        model = cases.make_structured(base, P=P, split_epoch=split, mig=float(rng.choice([0.2, 1.0, 4.0, 10.0])),
                                      sizes=rng.uniform(0.3, 2.0, P))
        if rng.random() < 0.3:
            model["mig_rates"][:, 1, 0] = 0.0
        mode = str(rng.choice(["plain", "bias", "vb", "plain"]))
        extra = {}
        if mode == "bias":
            extra.update(bias_heights=[float(rng.choice([200.0, 400.0, 3000.0]))], bias_strengths=[float(rng.choice([2.0, 5.0, 10.0])), 1.0],
                         delay_type=int(rng.integers(0, 3)), application_delays=np.full(E, float(rng.choice([0.5, 2000.0, 8000.0]))))
        if mode == "vb":
            extra.update(vb_coal_counts=rng.uniform(0.5, 50.0, (E, P)), vb_mig_counts=rng.uniform(0.5, 50.0, (E, P, P)))
        model = dict(model, **extra)
        segs = cases.make_segments(base, seed=int(rng.integers(1, 10**6)), unphased=bool(rng.integers(0, 2)),
                                   max_seg_len=int(rng.choice([2000, 5000])))
        run_seed = int(rng.integers(1, 10**6)); essf = float(rng.choice([0.0, 0.5, 0.9]))
        tag = "seed %d it %d: n %d E %d P %d split %d Np %d L %g %s ess %.1f" % (seed, it, n, E, P, split, Np, L, mode, essf)
        try:
            o = oracle.Oracle(model, Np, ess_fraction=essf, seed=run_seed, max_trace_events=32)
            o.init_prior(segs["start"][0]); o.run(o.pack_segments(model, segs))
        except RuntimeError as e:          # a capacity limit of the restatement (96 migration events per local tree)
            assert "too many migration events" in str(e), tag
            continue
        to, po, mo, co = o.trace(), o.particles(), o.migrations(), o.counts()
        for debug in (0, 2):
            g = ParticleFilter(model, Np, ess_fraction=essf, seed=run_seed, max_trace_events=32, debug=debug)
            g.init_prior(segs["start"][0]); g.load_segments(segs); g.run(); g.finish()
            tg, pg, mg, cg = g.trace(), g.particles(), g.migrations(), g.counts()
            assert (to["resampled"] == tg["resampled"]).all(), tag
            for k in ("T", "ess", "logl"):
                assert (_bits(to[k]) == _bits(tg[k])).all(), (tag, k)
            assert (po["children"] == pg["children"]).all(), tag
            for k in ("heights", "w_post", "w_pilot", "next_base"):
                assert (_bits(po[k]) == _bits(pg[k])).all(), (tag, k)
            assert (mo["n_events"] == mg["n_events"]).all() and (mo["node_pops"] == mg["node_pops"]).all(), tag
            for k in ("coal_count", "coal_opp", "rec_count", "rec_opp", "mig_count", "mig_opp"):
                np.testing.assert_allclose(cg[k], co[k], rtol=1e-8, atol=1e-8 * max(1e-300, np.abs(co[k]).max()), err_msg=tag + " " + k)
            g.close()
        o.close()
        done += 1
    assert done >= 5
