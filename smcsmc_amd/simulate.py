"""Synthetic .seg data: a small SMC' (sequentially Markov coalescent) simulator.

The reference generates its test data with the external `scrm` binary
(/root/reference/smcsmc/populationmodels.py:440-500) which is not available here, so this
module provides the data side of the benchmark: one population, piecewise-constant N(t),
n haplotypes, infinite-sites mutations, written with the conventions of
populationmodels.convert_scrm_to_seg (populationmodels.py:502-577): integer positions
int(x*L+0.5), a leading position 1, a trailing all-missing row.

It is deliberately independent of both the HIP path and the oracle (numpy RNG, its own tree
code): it only has to produce realistic input.
"""
import numpy as np


def default_epochs(n_epochs=32, t_first=133.0, t_last=133032.0):
    """Log-spaced epoch boundaries in generations, like the front-end's -P 133 133032 31*1."""
    if n_epochs == 1:
        return np.array([0.0])
    inner = np.exp(np.linspace(np.log(t_first), np.log(t_last), n_epochs - 1))
    return np.concatenate([[0.0], inner])


class _Tree:
    def __init__(self, n):
        self.n = n
        self.parent = -np.ones(2 * n - 1, dtype=np.int64)
        self.height = np.zeros(2 * n - 1)
        self.children = -np.ones((2 * n - 1, 2), dtype=np.int64)
        self.root = -1


def _coal_time(rng, t, lineages_at, change_times, pop_sizes):
    """Waiting time for a lineage starting at t to coalesce, rate k(t)/(2N(t)); lineages_at(t) -> (k, t_next)."""
    E = len(change_times)
    e = int(np.searchsorted(change_times, t, side="right") - 1)
    x = rng.exponential()
    while True:
        k, tn_node = lineages_at(t)
        tn_ep = change_times[e + 1] if e + 1 < E else np.inf
        tn = min(tn_node, tn_ep)
        rate = k / (2.0 * pop_sizes[e])
        if x > (tn - t) * rate:
            x -= (tn - t) * rate
            t = tn
            if tn_ep <= tn:
                e += 1
            continue
        return t + x / rate


def simulate_seg(n, L, mu, rho, change_times, pop_sizes, seed=1, missing=()):
    """Returns dict(start, length, alleles[n_rows, n]) with 1-based integer starts (file coordinates)."""
    rng = np.random.default_rng(seed)
    change_times = np.asarray(change_times, float)
    pop_sizes = np.asarray(pop_sizes, float)
    # ---- initial tree: Kingman coalescent with piecewise-constant N ----
    heights = np.zeros(2 * n - 1)
    parent = -np.ones(2 * n - 1, dtype=np.int64)
    active = list(range(n))
    t = 0.0
    nxt = n
    while len(active) > 1:
        k = len(active)
        # pairwise rate k(k-1)/2 / (2N)
        E = len(change_times)
        e = int(np.searchsorted(change_times, t, side="right") - 1)
        x = rng.exponential()
        while True:
            tn = change_times[e + 1] if e + 1 < E else np.inf
            rate = k * (k - 1) / 2.0 / (2.0 * pop_sizes[e])
            if x > (tn - t) * rate:
                x -= (tn - t) * rate
                t = tn
                e += 1
                continue
            t = t + x / rate
            break
        i, j = rng.choice(len(active), 2, replace=False)
        a, b = active[i], active[j]
        heights[nxt] = t
        parent[a] = nxt
        parent[b] = nxt
        active = [v for idx, v in enumerate(active) if idx not in (i, j)] + [nxt]
        nxt += 1
    root = active[0]

    def branch_table():
        ids = [v for v in range(2 * n - 1) if v != root and (parent[v] >= 0)]
        lens = np.array([heights[parent[v]] - heights[v] for v in ids])
        return ids, lens

    def leaves_under(v):
        if v < n:
            return [v]
        kids = [c for c in range(2 * n - 1) if parent[c] == v]
        out = []
        for c in kids:
            out += leaves_under(c)
        return out

    positions = []
    patterns = []
    x = 0.0
    while x < L:
        ids, lens = branch_table()
        ltree = lens.sum()
        dx = rng.exponential(1.0 / (rho * ltree)) if rho > 0 else np.inf
        x_next = min(L, x + dx)
        # mutations on this stretch
        nm = rng.poisson(mu * ltree * (x_next - x))
        if nm:
            mpos = np.sort(rng.uniform(x, x_next, nm))
            mbr = rng.choice(len(ids), nm, p=lens / ltree)
            for ppos, b in zip(mpos, mbr):
                pat = np.zeros(n, dtype=np.int8)
                pat[leaves_under(ids[b])] = 1
                positions.append(ppos)
                patterns.append(pat)
        x = x_next
        if x >= L:
            break
        # ---- SMC' transition: cut a uniform point, re-coalesce (the cut branch stays a target) ----
        b = ids[rng.choice(len(ids), p=lens / ltree)]
        h = rng.uniform(heights[b], heights[parent[b]])
        p_old = parent[b]

        def lineages_at(tt):
            k = 0
            tn = np.inf
            for v in range(2 * n - 1):
                if v == root:
                    if heights[v] <= tt:
                        k += 1
                    elif heights[v] < tn:
                        tn = heights[v]
                    continue
                if parent[v] < 0:
                    continue
                if heights[v] <= tt < heights[parent[v]]:
                    k += 1
                if heights[v] > tt and heights[v] < tn:
                    tn = heights[v]
                if heights[parent[v]] > tt and heights[parent[v]] < tn:
                    tn = heights[parent[v]]
            return k, tn

        tc = _coal_time(rng, h, lineages_at, change_times, pop_sizes)
        cands = []
        for v in range(2 * n - 1):
            if v == root:
                if heights[v] <= tc:
                    cands.append(v)
            elif parent[v] >= 0 and heights[v] <= tc < heights[parent[v]]:
                cands.append(v)
        c = cands[rng.integers(len(cands))]
        if c == b:
            continue      # coalesced back into its own branch: tree unchanged
        sib = [v for v in range(2 * n - 1) if parent[v] == p_old and v != b][0]
        g = parent[p_old]
        if c == p_old:
            c = sib
        # detach p_old
        parent[sib] = g
        if p_old == root:
            root = sib
            parent[sib] = -1
        # re-use node p_old as the new coalescence node
        pc = parent[c] if c != root else -1
        heights[p_old] = tc
        parent[p_old] = pc
        if c == root:
            root = p_old
            parent[p_old] = -1
        parent[c] = p_old
        parent[b] = p_old

    return sites_to_seg(positions, patterns, n, L, missing)


def sites_to_seg(positions, patterns, n, L, missing=()):
    """.seg rows from ascending site positions and 0/1 patterns, with the conventions of convert_scrm_to_seg
    (populationmodels.py:502-577): integer positions int(x+0.5), a leading position 1, a trailing all-missing row."""
    ipos = [1]
    rows = []
    for ppos, pat in zip(positions, patterns):
        ip = int(ppos + 0.5)
        if ip <= ipos[-1]:
            continue        # two mutations rounded onto the same base: keep the first
        ipos.append(ip)
        rows.append(pat)
    start, length, alleles = [], [], []
    for idx in range(len(ipos) - 1):
        start.append(ipos[idx])
        length.append(ipos[idx + 1] - ipos[idx])
        a = np.array(rows[idx], np.int8).copy()
        for m in missing:
            a[m] = -1
        alleles.append(a)
    start.append(ipos[-1])
    length.append(int(L + 0.5) - ipos[-1])
    alleles.append(-np.ones(n, dtype=np.int8))
    return {"start": np.array(start, np.int64), "length": np.array(length, np.int64),
            "alleles": np.array(alleles, np.int8).reshape(-1, n)}


def simulate_seg_device(n, L, mu, rho, change_times, pop_sizes, seed=1, nchunks=1, device=0, missing=(), structure=None):
    """The same data model simulated on the GPU (k_simulate: one lane per chunk, the filter's own SMC' transition,
    Philox stream 3): a list of `nchunks` independent chunks in the format of simulate_seg.  A 100 Mb chunk of two
    diploids takes well under a second; the chunks of a call run side by side.  `structure` = the structured part of a
    model dictionary (n_pops, pop_sizes [E][P], mig_rates, single_mig, sample_pops): the data then comes from that
    isolation-with-migration model (k_simulate_mp)."""
    from . import pf
    change_times = np.asarray(change_times, float)
    model = dict(change_times=change_times, pop_sizes=np.asarray(pop_sizes, float), lags=np.ones(len(change_times)), nsam=int(n),
                 loci_length=float(L), mutation_rate=float(mu), recombination_rate=float(rho))
    if structure:
        model.update({k: structure[k] for k in ("n_pops", "pop_sizes", "mig_rates", "single_mig", "sample_pops") if k in structure})
    out = []
    for pos, masks in pf.simulate_sites(model, seed=seed, nchunks=nchunks, device=device):
        pats = ((masks[:, None] >> np.arange(n)[None, :]) & 1).astype(np.int8)
        out.append(sites_to_seg(pos, pats, n, L, missing))
    return out


def write_seg(path, seg):
    code = {-1: ".", 0: "0", 1: "1", 2: "/"}
    with open(path, "w") as f:
        for s, l, a in zip(seg["start"], seg["length"], seg["alleles"]):
            f.write("%d\t%d\tT\tF\t1\t%s\n" % (s, l, "".join(code[int(v)] for v in a)))
