"""Shared test inputs: models and synthetic segment sets (seeded, small enough for the oracle)."""
import numpy as np

from smcsmc_amd import segments as segmod
from smcsmc_amd import simulate


def make_model(n=4, E=8, L=2e5, N0=1e4, mu=2.5e-8, rho=1e-8, lag=None, sizes=None, **kw):
    ct = simulate.default_epochs(E, 133.0, 133032.0) if E > 1 else np.array([0.0])
    ps = np.full(E, N0) if sizes is None else np.asarray(sizes, float) * N0
    if lag is None:
        # the reference's uncalibrated default (count.cpp:230-247): 4 / (rho * top_t)
        lags = [20000.0] if E == 1 else [4.0 / (rho * (ct[e + 1] if e + 1 < E else ct[-1])) for e in range(E)]
    else:
        lags = [float(lag)] * E
    m = dict(change_times=ct, pop_sizes=ps, lags=np.array(lags), nsam=n, loci_length=float(L),
             mutation_rate=mu, recombination_rate=rho)
    m.update(kw)
    return m


def make_segments(model, seed=1, unphased=False, missing_block=None, max_seg_len=None, tmpdir=None):
    """Simulated data packed exactly as the host side packs a .seg file."""
    n = model["nsam"]
    L = model["loci_length"]
    seg = simulate.simulate_seg(n, L, model["mutation_rate"], model["recombination_rate"],
                                model["change_times"], model["pop_sizes"], seed=seed)
    al = seg["alleles"].copy()
    if unphased:
        for i in range(0, n - 1, 2):
            het = (al[:, i] >= 0) & (al[:, i + 1] >= 0) & (al[:, i] != al[:, i + 1])
            al[het, i] = 2
            al[het, i + 1] = 2
    if missing_block is not None:
        a, b, cols = missing_block
        rows = (seg["start"] >= a) & (seg["start"] < b)
        for c in cols:
            al[rows, c] = -1
    seg["alleles"] = al
    S = segmod.Segments.from_sites(seg["start"], seg["length"], seg["alleles"], n, L, max_segment_length=max_seg_len or 1e99)
    return S.pack(model["lags"])


def nodata_segments(model, seglen=1000.0):
    L = model["loci_length"]; n = model["nsam"]
    S = int(np.ceil(L / seglen))
    return dict(start=np.arange(S) * seglen, length=np.full(S, seglen), state=np.full(S, 1, np.int8),
                alleles=np.full((S, n), -1, np.int8),
                max_record_epoch=np.full(S, -1, np.int32) * 0 + (len(model["lags"]) - 1))


def make_structured(model, P=2, split_epoch=None, mig=1.0, N0=1e4, sample_pops=None, sizes=None):
    """Turns a single-population model into an isolation-with-migration model with P populations:
    symmetric migration 4*N0*m = `mig` between all pairs until the epoch `split_epoch`, at whose start every
    population joins population 0 (scrm -ej).  Shape of the reference's test_two_pops.py:54-72."""
    m = dict(model)
    E = len(m["change_times"])
    n = m["nsam"]
    split_epoch = E - 2 if split_epoch is None else split_epoch
    base = np.asarray(m["pop_sizes"], float).reshape(E)
    ps = np.repeat(base[:, None], P, axis=1)
    if sizes is not None:
        ps = ps * np.asarray(sizes, float)[None, :]
    mr = np.zeros((E, P, P)); sm = np.zeros((E, P, P))
    for e in range(E):
        if e < split_epoch:
            for a in range(P):
                for b in range(P):
                    if a != b:
                        mr[e, a, b] = mig / (4.0 * N0)
    if split_epoch < E:
        for a in range(1, P):
            sm[split_epoch, a, 0] = 1.0
    m.update(n_pops=P, pop_sizes=ps, mig_rates=mr, single_mig=sm,
             sample_pops=list(sample_pops) if sample_pops is not None else [i * P // n for i in range(n)])
    return m
