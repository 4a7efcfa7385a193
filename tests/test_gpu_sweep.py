"""The multi-chunk row pipeline (k_sweep, pf_run_many): the argument block read from device memory, the count windows and
the plan of every step worked out on the device, several chunks per launch.  Every chunk must be bit-identical to its
own single-chunk run through k_pipe (argument block by value, windows from the host) and to the oracle."""
import numpy as np
import pytest

import cases
from smcsmc_amd import ParticleFilter

pytestmark = pytest.mark.gpu
K_PIPE = 16          # PF_DEBUG_K_PIPE


def _bits(x):
    return np.asarray(x, np.float64).view(np.uint64)


def _run_alone(model, segs, Np, seed, debug, step=None, **kw):
    f = ParticleFilter(model, Np, seed=seed, debug=debug, **kw)
    f.init_prior(0.0); f.load_segments(segs)
    if step is None:
        f.run()
    else:
        n = len(segs["start"])
        for s0 in range(0, n, step):
            f.run(s0, min(n, s0 + step))
    f.finish()
    return f




def _same(a, b, counts_rtol=None):
    """trees, weights, traces and resampling bit for bit; the counts too, unless the two runs group their sums differently
    (counting by generation against counting by epoch: the same terms, rounding-level differences)"""
    assert _bits(a.logl()) == _bits(b.logl())
    ta, tb = a.trace(), b.trace()
    for k in ("T", "ess", "logl"):
        assert (_bits(ta[k]) == _bits(tb[k])).all(), k
    assert (ta["resampled"] == tb["resampled"]).all()
    sa, pa = a.resample_events(); sb, pb = b.resample_events()
    assert (sa == sb).all() and (pa == pb).all()
    ca, cb = a.counts(), b.counts()
    for k in ca:
        if counts_rtol is None or k in ("resample_count", "logl", "delayed_opp", "delayed_count"):
            assert (_bits(ca[k]) == _bits(cb[k])).all(), k
        else:
            np.testing.assert_allclose(ca[k], cb[k], rtol=counts_rtol, atol=1e-300, err_msg=k)
    wa, wb = a.particles(), b.particles()
    for k in wa:
        assert (np.asarray(wa[k]).view(np.uint8) == np.asarray(wb[k]).view(np.uint8)).all(), k


@pytest.mark.parametrize("n,Np,biased", [(4, 1000, False), (2, 300, False), (8, 700, False), (4, 640, True), (6, 513, True)])
def test_sweep_equals_k_pipe_and_oracle(oracle, hiplib, n, Np, biased):
    import oracle_lib
    model = cases.make_model(n=n, E=10, L=1.5e5)
    if biased:
        model.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0], delay_type=0, application_delays=np.full(10, 3000.0))
    segs = cases.make_segments(model, seed=11, max_seg_len=4000)
    a = _run_alone(model, segs, Np, 5, 0, local_recomb=True)               # k_sweep, one chunk
    b = _run_alone(model, segs, Np, 5, K_PIPE, local_recomb=True)          # k_pipe
    _same(a, b)
    la, lb = a.local_recomb(), b.local_recomb()
    for k in la:                                                            # atomics: same terms, any order
        np.testing.assert_allclose(la[k], lb[k], rtol=1e-9, atol=1e-9 * max(1e-300, float(np.abs(lb[k]).max())))
    o = oracle_lib.Oracle(model, Np, seed=5)
    o.init_prior(0.0); o.run(o.pack_segments(model, segs))
    assert _bits(a.logl()) == _bits(o.logl())
    sg, pg = a.resample_events(); so, po = o.resample_events()
    assert (sg == so).all() and (pg == po).all()
    co, cg = o.counts(), a.counts()
    for k in ("coal_count", "coal_opp", "rec_count", "rec_opp"):
        np.testing.assert_allclose(cg[k], co[k], rtol=1e-9, atol=1e-300)


NO_DRAW_TABLE = 128  # PF_DEBUG_NO_DRAW_TABLE


@pytest.mark.parametrize("n,biased,mu,rho,step", [(4, False, 2.5e-8, 1e-8, None), (4, False, 2.5e-9, 4e-8, None), (8, False, 2.5e-9, 2e-8, 23),
                                                 (4, True, 2.5e-8, 1e-8, 31), (6, True, 4e-9, 3e-8, None)])
def test_draw_table_changes_nothing(oracle, hiplib, n, biased, mu, rho, step):
    """The random numbers made ahead by the draw role of k_sweep are the ones an update would compute itself: with and
    without the table, trees, weights, resampling and counts are the same bits, and equal the oracle's.  The sparse-data
    cases make a dozen or more recombinations per row and particle, so slots outrun their sixteen updates' worth of table,
    fall back to their own numbers and let the table skip ahead; `step` cuts the sweep into calls (the table restarts)."""
    import oracle_lib
    model = cases.make_model(n=n, E=8, L=4e5, mu=mu, rho=rho)
    if biased:
        model.update(bias_heights=[400.0], bias_strengths=[3.0, 1.0], delay_type=0, application_delays=np.full(8, 3000.0))
    segs = cases.make_segments(model, seed=4, max_seg_len=30000)
    a = _run_alone(model, segs, 600, 3, 0, step=step)
    b = _run_alone(model, segs, 600, 3, NO_DRAW_TABLE)
    _same(a, b)
    if rho > 1e-8:
        per_row = float(np.sum(a.counts()["rec_count"])) / len(segs["start"])
        assert per_row > 5.0, per_row                      # with 600 particles the table (sixteen updates per row) is outrun in most rows
    o = oracle_lib.Oracle(model, 600, seed=3)
    o.init_prior(0.0); o.run(o.pack_segments(model, segs))
    assert _bits(a.logl()) == _bits(o.logl())
    co, cg = o.counts(), a.counts()
    for k in ("coal_count", "coal_opp", "rec_count", "rec_opp"):
        np.testing.assert_allclose(cg[k], co[k], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("debug", [64, 64 + 1024, 512, 256, 8192, 1 << 23, 4 << 16], ids=["split-roles", "split-roles-cu-masks", "count-young-first", "four-way-search", "flag-handoff", "one-launch", "second-launches-in-fours"])
def test_launch_arrangements_change_nothing(hiplib, debug):
    """The A/B switches of the row pipeline -- the extend role and the other roles as two launches on two streams (with the
    draw role riding with the extend launch), the two streams on disjoint compute units, the count columns in ascending
    epoch order, the four-way epoch search -- give the same bits as the default single launch."""
    model = cases.make_model(n=4, E=12, L=1.5e5)
    segs = cases.make_segments(model, seed=9, max_seg_len=3500)
    a = _run_alone(model, segs, 1100, 4, 0)
    b = _run_alone(model, segs, 1100, 4, debug, step=(41 if debug in (64, 8192, 1 << 23, 4 << 16) else None))
    _same(a, b)


def test_sweep_in_pieces_equals_one_call(hiplib):
    """pf_run over [0, S) in calls of 37 rows (two flush steps and a fresh window seed each) = one call."""
    model = cases.make_model(n=4, E=8, L=1.2e5)
    segs = cases.make_segments(model, seed=3, max_seg_len=3000)
    a = _run_alone(model, segs, 900, 2, 0)
    b = _run_alone(model, segs, 900, 2, 0, step=37)
    _same(a, b)


def test_chunks_in_one_launch_equal_their_own_runs(hiplib):
    """Five chunks with different data, lengths and seeds through pf_run_many: each equals its pf_run, bit for bit."""
    model = cases.make_model(n=4, E=12, L=2e5)
    chunks = []
    for k in range(5):
        m = dict(model, loci_length=float(model["loci_length"] * (0.5 + 0.125 * k)))
        segs = cases.make_segments(m, seed=20 + k, max_seg_len=5000)
        chunks.append((m, segs))
    alone = [_run_alone(m, sg, 1024, 7 + k, K_PIPE) for k, (m, sg) in enumerate(chunks)]
    many = []
    for k, (m, sg) in enumerate(chunks):
        f = ParticleFilter(m, 1024, seed=7 + k)
        f.init_prior(0.0); f.load_segments(sg)
        many.append(f)
    nmax = max(f.n_segs for f in many)
    assert len({f.n_segs for f in many}) > 1                # the chunks do not end together
    for s0 in range(0, nmax, 300):                          # several calls: chunks that are done sit the later ones out
        ParticleFilter.run_many(many, s0, min(nmax, s0 + 300))
    for f in many:
        f.finish()
    for f, g in zip(many, alone):
        assert f.segments_done() == g.segments_done()
        _same(f, g)


@pytest.mark.parametrize("workers,count_wgs", [(7, 3), (40, 0), (1, 2)])
def test_count_workers_change_nothing(hiplib, workers, count_wgs):
    """pf_params.count_workers: the ledger and count work of a step taken off a queue by a fixed number of workgroups instead of one
    workgroup per item in the launch -- every item writes its own accumulators, so the bits are those of the static form, for one chunk
    and for chunks that run in one launch (and end at different rows)."""
    model = cases.make_model(n=4, E=12, L=2e5)
    chunks = []
    for k in range(3):
        m = dict(model, loci_length=float(model["loci_length"] * (0.6 + 0.2 * k)))
        chunks.append((m, cases.make_segments(m, seed=30 + k, max_seg_len=4000)))
    static = [_run_alone(m, sg, 1100, 5 + k, 0, count_wgs=count_wgs, local_recomb=True) for k, (m, sg) in enumerate(chunks)]
    alone = _run_alone(chunks[0][0], chunks[0][1], 1100, 5, 0, count_wgs=count_wgs, count_workers=workers, local_recomb=True, step=53)
    _same(alone, static[0])
    many = []
    for k, (m, sg) in enumerate(chunks):
        f = ParticleFilter(m, 1100, seed=5 + k, count_wgs=count_wgs, count_workers=workers, local_recomb=True)
        f.init_prior(0.0); f.load_segments(sg)
        many.append(f)
    nmax = max(f.n_segs for f in many)
    for s0 in range(0, nmax, 211):
        ParticleFilter.run_many(many, s0, min(nmax, s0 + 211))
    for f, g in zip(many, static):
        f.finish()
        _same(f, g)
        for key in ("opp_diff", "counts"):              # (added with atomics in either form: equal to rounding, bins that cancel to nothing included)
            ref = g.local_recomb()[key]
            np.testing.assert_allclose(f.local_recomb()[key], ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
    assert sum(int(f.trace()["resampled"].sum()) for f in many) > 10        # the ledger items were there to be taken


def test_chunks_as_two_launches_per_step_equal_their_own_single_launch_runs(hiplib):
    """The default for the headline shape: the extend, bookkeeping and draw roles of all chunks as one launch per step, their ledger and
    count roles as a second launch on the counting stream (four workgroups to a compute unit), paced by the sixteen-slot ring.  Every
    chunk is bit-identical to its own run with everything in ONE launch (PF_DEBUG_ONE_LAUNCH), in several calls, with chunks that end
    at different rows."""
    model = cases.make_model(n=4, E=12, L=2e5)
    chunks = []
    for k in range(4):
        m = dict(model, loci_length=float(model["loci_length"] * (0.55 + 0.15 * k)))
        chunks.append((m, cases.make_segments(m, seed=40 + k, max_seg_len=4000)))
    alone = [_run_alone(m, sg, 1100, 9 + k, 1 << 23, count_wgs=3, local_recomb=True) for k, (m, sg) in enumerate(chunks)]
    many = []
    for k, (m, sg) in enumerate(chunks):
        f = ParticleFilter(m, 1100, seed=9 + k, count_wgs=3, local_recomb=True)
        f.init_prior(0.0); f.load_segments(sg)
        many.append(f)
    nmax = max(f.n_segs for f in many)
    for s0 in range(0, nmax, 173):
        ParticleFilter.run_many(many, s0, min(nmax, s0 + 173))
    for f, g in zip(many, alone):
        f.finish()
        _same(f, g)
        for key in ("opp_diff", "counts"):
            ref = g.local_recomb()[key]
            np.testing.assert_allclose(f.local_recomb()[key], ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


def test_run_many_rejects_chunks_of_different_shape(hiplib):
    from smcsmc_amd import PfError
    m1 = cases.make_model(n=4, E=8, L=5e4)
    s1 = cases.make_segments(m1, seed=1, max_seg_len=3000)
    a = ParticleFilter(m1, 512, seed=1); a.init_prior(0.0); a.load_segments(s1)
    b = ParticleFilter(m1, 256, seed=1); b.init_prior(0.0); b.load_segments(s1)
    with pytest.raises(PfError, match="must share"):
        ParticleFilter.run_many([a, b])
    with pytest.raises(PfError, match="twice"):
        ParticleFilter.run_many([a, a])


@pytest.mark.parametrize("n,P,Np,biased", [(8, 2, 640, False), (4, 2, 1000, False), (6, 3, 500, False), (4, 2, 512, True)])
def test_structured_rows_on_the_pipeline_equal_the_two_stream_path(hiplib, n, P, Np, biased):
    """Structured models (register-tree kernel): the row pipeline (extend launches with the decision in their prologue on
    the filter stream, bookkeeping / ledger / counts as their own launches on the counting stream) against the round-2
    path (k_extend_mpr + k_decide, PF_DEBUG_K_PIPE) -- trees, weights, migration events, resampling indices bit for bit,
    the lagged counts too (same workgroups, same order); in one call and in calls of 23 rows."""
    base = cases.make_model(n=n, E=8, L=1.2e5)
    segs = cases.make_segments(base, seed=31 + n, max_seg_len=4000)
    model = cases.make_structured(base, P=P, split_epoch=5, mig=1.5)
    if biased:
        model = dict(model, bias_heights=[400.0], bias_strengths=[4.0, 1.0], application_delays=np.full(8, 2500.0), delay_type=0)
    a = _run_alone(model, segs, Np, 9, 0, local_recomb=True)
    b = _run_alone(model, segs, Np, 9, K_PIPE, local_recomb=True)
    c = _run_alone(model, segs, Np, 9, 0, step=23, local_recomb=True)
    for x in (b, c):
        _same(a, x)
        ma, mx = a.migrations(), x.migrations()
        for k in ma:
            assert (np.asarray(ma[k]) == np.asarray(mx[k])).all(), k
