"""Exact E-step of the two-sample SMC' model by forward-backward on a discretised coalescence time.

TEST INFRASTRUCTURE (numpy only).  Nothing here comes from the reference or from oracle/: it is an independent
statement of the *model* the particle filter samples from, used to check that the filter (oracle and HIP path alike)
converges to the exact posterior expectations as the number of particles grows.

Model (n = 2, one population, piecewise-constant N(t); times in generations, positions in bp):
  * hidden state along the sequence: the coalescence time s of the two samples; prior density
    pi(s) = lam(s) exp(-Lam(s)), lam = 1/(2N), Lam = int_0^s lam;
  * recombinations arrive at rate rho * 2s per bp; the cut height h is uniform on (0, s) (either branch);
    the floating lineage re-coalesces upwards at rate 2 lam(t) for h < t < s (the other branch *and* the stub of
    its own branch: SMC') and at rate lam(t) above s (pairwise with the old root lineage);
    below s the target is the stub with probability 1/2 (tree unchanged), otherwise the new coalescence time is t;
  * emission: no mutation over a row of length d on both branches, exp(-mu 2s d); at the row's last base a
    heterozygous site p(1-p), a homozygous one (p^2 + (1-p)^2)/2, p = exp(-mu s); rows with a missing sample carry
    no information (factor 1/2 at the site, none along the row).

Sufficient statistics (what the reference's CountModel accumulates; src/count.cpp:495-555, src/particle.cpp:193-390):
  * recombination opportunity of epoch e: 2 |[0,s] n e| per bp; recombination count: the epoch that holds h;
  * coalescence opportunity of epoch e, per recombination: 2 |[h, min(t,s)] n e| + |[s, t] n e| (the number of
    lineages the floating lineage can join, integrated over the time it floats); coalescence count: the epoch that
    holds t -- also when it re-joins its own stub; the initial tree adds |[0,s] n e| and one event at x = 0.

Method.  The jump flux of SMC' between coalescence times is f(lo, hi) = 2 rho * [lam(lo) A(lo)] * [lam(hi) exp(-Lam(hi))]
with A(t) = int_0^t exp(-2(Lam(t) - Lam(h))) dh -- symmetric (the process is reversible) and *separable*, so cell-to-cell
rates and every flux-weighted functional above reduce to one-dimensional tables over the cells, integrated by
Gauss-Legendre with closed forms of Lam, A and their integrals per epoch.  Cells are aligned with the epoch starts.
Within a row the posterior evolves by exp(G d) with G = (jump rates) - diag(mutation killing); G is symmetrised with the
prior and diagonalised once per missing-data class; expected occupation and jump counts are the usual integrals of
exp(Gx) (.) exp(G(d-x)) in the eigenbasis, accumulated over rows and transformed back once.

Discretisation error: O(cell width^2); `refine()` below reports the change under doubling of the grid.
"""
import numpy as np

_GLX, _GLW = np.polynomial.legendre.leggauss(10)


class Grid:
    def __init__(self, change_times, ne, K=400, t0=30.0, lam_max=34.0):
        self.T = np.asarray(change_times, float)
        self.lam = 1.0 / (2.0 * np.asarray(ne, float))
        E = self.E = len(self.T)
        assert self.T[0] == 0.0
        # cumulative intensity and A(t) = exp(-2 Lam) int_0^t exp(2 Lam) at the epoch starts
        self.LamT = np.zeros(E)
        self.AT = np.zeros(E)
        for e in range(1, E):
            w = self.T[e] - self.T[e - 1]
            l = self.lam[e - 1]
            self.LamT[e] = self.LamT[e - 1] + l * w
            self.AT[e] = 1 / (2 * l) + (self.AT[e - 1] - 1 / (2 * l)) * np.exp(-2 * l * w)
        self.I0T = np.exp(2 * self.LamT) * self.AT
        tmax = self.T[-1] + (lam_max - self.LamT[-1]) / self.lam[-1]
        assert tmax > self.T[-1]
        # cell edges: uniform in log(t + t0) inside every epoch, the number of cells in proportion to its share
        bounds = np.append(self.T, tmax)
        u = np.log(bounds + t0)
        share = np.diff(u) / (u[-1] - u[0])
        edges = [0.0]
        for e in range(E):
            k = max(2, int(round(share[e] * K)))
            ue = np.linspace(u[e], u[e + 1], k + 1)[1:]
            edges.extend(list(np.exp(ue) - t0))
            edges[-1] = bounds[e + 1]
        self.edges = np.array(edges)
        self.K = len(self.edges) - 1
        self.lo, self.hi = self.edges[:-1], self.edges[1:]
        self.w = self.hi - self.lo
        self.ep = np.searchsorted(self.T, 0.5 * (self.lo + self.hi), side="right") - 1
        self._tables()

    # ---- closed forms (vectorised over t)
    def epoch_of(self, t):
        return np.clip(np.searchsorted(self.T, t, side="right") - 1, 0, self.E - 1)

    def Lam(self, t):
        e = self.epoch_of(t)
        return self.LamT[e] + self.lam[e] * (t - self.T[e])

    def A(self, t):
        e = self.epoch_of(t)
        l = self.lam[e]
        return 1 / (2 * l) + (self.AT[e] - 1 / (2 * l)) * np.exp(-2 * l * (t - self.T[e]))

    def Le(self, t, e):
        """|[0,t] n epoch e|"""
        top = self.T[e + 1] if e + 1 < self.E else np.inf
        return np.clip(np.minimum(t, top) - self.T[e], 0.0, None)

    def I0e(self, t, e):
        """int_{[0,t] n e} exp(2 Lam(h)) dh"""
        l = self.lam[e]
        return np.exp(2 * self.LamT[e]) * np.expm1(2 * l * self.Le(t, e)) / (2 * l)

    def Je(self, t, e):
        """int_{[0,t] n e} I0(tau) dtau,  I0(tau) = exp(2 Lam(tau)) A(tau) = int_0^tau exp(2 Lam)"""
        l = self.lam[e]
        x = self.Le(t, e)
        c = np.exp(2 * self.LamT[e])
        return self.I0T[e] * x + c * (np.expm1(2 * l * x) - 2 * l * x) / (4 * l * l)

    def _gl(self, f):
        """integral of f over every cell (f vectorised over an array of shape [K, nq])"""
        half = 0.5 * self.w[:, None]
        t = 0.5 * (self.lo + self.hi)[:, None] + half * _GLX[None, :]
        return (f(t) * _GLW[None, :]).sum(1) * half[:, 0]

    def _tables(self):
        E, K = self.E, self.K
        lamc = self.lam[self.ep]
        eL_lo, eL_hi = np.exp(-self.Lam(self.lo)), np.exp(-self.Lam(self.hi))
        self.pi = eL_lo - eL_hi
        self.pi[-1] = eL_lo[-1]                       # the tail beyond the last edge is lumped into the last cell
        dens = lambda t: self.lam[self.epoch_of(t)] * np.exp(-self.Lam(t))
        self.sbar = self._gl(lambda t: t * dens(t)) / self._gl(dens)
        lamA = lambda t: self.lam[self.epoch_of(t)] * self.A(t)
        self.la = 0.5 * (self.w - (self.A(self.hi) - self.A(self.lo)))        # int lam A = (t - A)/2, from A' = 1 - 2 lam A
        self.lat = self._gl(lambda t: lamA(t) * (self.hi[:, None] - t))
        self.pt = self._gl(lambda t: dens(t) * (t - self.lo[:, None]))
        self.d = np.zeros((K, E))      # int_cell lam(t) exp(-2Lam(t)) 2 J_e(t) dt: opportunity, events from below the old root
        self.r = np.zeros((K, E))      # int_cell lam(t) exp(-2Lam(t)) I0e(t) dt: the epoch of the cut height
        for e in range(E):
            self.d[:, e] = self._gl(lambda t: self.lam[self.epoch_of(t)] * np.exp(-2 * self.Lam(t)) * 2 * self.Je(t, e))
            self.r[:, e] = self._gl(lambda t: self.lam[self.epoch_of(t)] * np.exp(-2 * self.Lam(t)) * self.I0e(t, e))
        LA = lambda t: 0.5 * (t - self.A(t))
        # flux of events whose new time lies in the cell of the old one, per unit 2 rho: below (stub or other branch), above
        self.sw = self._gl(lambda t: dens(t) * (LA(t) - LA(self.lo)[:, None]))
        self.uw = self._gl(lambda t: lamA(t) * (np.exp(-self.Lam(t)) - eL_hi[:, None]))
        self.uw[-1] = self._gl(lambda t: lamA(t) * np.exp(-self.Lam(t)))[-1]
        _ = lamc

    def check(self):
        """identities the tables must satisfy (returns the largest relative violations)"""
        cum_la = np.concatenate([[0.0], np.cumsum(self.la)])[:-1]
        tail_pi = self.pi[::-1].cumsum()[::-1] - self.pi
        total = 2 * cum_la * self.pi + 2 * self.sw + self.uw + self.la * tail_pi
        v1 = np.abs(total / (self.pi * self.sbar) - 1).max()          # all events of a state: rate 2 rho s
        v2 = np.abs(self.r.sum(1) / self.la - 1).max()                # the cut height lies in some epoch
        v3 = np.abs(self.sw / self.uw - 1)[:-1].max()                 # reversibility inside a cell
        return v1, v2, v3


class ExactHMM2:
    """run(rows) -> dict(logl, coal_count[E], coal_opp[E], rec_count[E], rec_opp[E]).

    rows: dict with start, length (bp), state (0 = site at the end of the row), alleles [S,2] in {-1,0,1,2},
    max_record_epoch [S] (events of a row are recorded in epochs <= this; src/smcsmc.cpp:266-275)."""

    def __init__(self, change_times, ne, mu, rho, K=400, **kw):
        self.g = Grid(change_times, ne, K=K, **kw)
        self.mu, self.rho = float(mu), float(rho)
        g = self.g
        K = g.K
        r2 = 2 * self.rho
        sq = np.sqrt(g.pi)
        # symmetrised jump rates between cells: S_ab = Q_ab sqrt(pi_a / pi_b) = 2 rho la_lo sqrt(pi_hi / pi_lo)
        S = r2 * np.triu(np.outer(g.la / sq, sq), 1)
        self.S = S + S.T
        Q = self.S * (sq[None, :] / sq[:, None])
        self.out = Q.sum(1)
        self.eig = {}
        for cls, kill in ((1, self.mu * 2 * g.sbar), (0, np.zeros(K))):
            d, U = np.linalg.eigh(self.S - np.diag(self.out + kill))
            dd = d[:, None] - d[None, :]
            np.fill_diagonal(dd, 1.0)
            self.eig[cls] = (d, U, 1.0 / dd)
        p = np.exp(-self.mu * g.sbar)
        self.em = {"het": p * (1 - p), "hom": 0.5 * (p * p + (1 - p) * (1 - p)), "half": np.full(K, 0.5), "one": np.ones(K)}

    def _row_classes(self, rows):
        al = np.asarray(rows["alleles"]).reshape(len(rows["start"]), -1)
        assert al.shape[1] == 2
        miss = (al < 0).sum(1)
        kill_cls = (miss == 0).astype(int)
        em = []
        for s in range(len(al)):
            if rows["state"][s] != 0: em.append("one")
            elif miss[s] == 2: em.append("one")
            elif miss[s] == 1: em.append("half")
            elif al[s, 0] == 2 or al[s, 1] == 2 or al[s, 0] != al[s, 1]: em.append("het")
            else: em.append("hom")
        return kill_cls, em

    def run(self, rows, seq_len=None):
        g = self.g
        K, E = g.K, g.E
        start = np.asarray(rows["start"], float)
        length = np.asarray(rows["length"], float)
        if seq_len is not None:
            length = np.minimum(start + length, seq_len) - np.minimum(start, seq_len)
        nrow = len(start)
        limit = np.asarray(rows.get("max_record_epoch", np.full(nrow, E - 1)), int)
        kill_cls, em = self._row_classes(rows)
        sq = np.sqrt(g.pi)
        # forward
        at = np.empty((nrow + 1, K))
        at[0] = sq                                   # alpha~ = alpha / sqrt(pi), alpha_0 = pi
        c = np.empty(nrow)
        for s in range(nrow):
            d, U, _ = self.eig[kill_cls[s]]
            v = ((at[s] @ U) * np.exp(d * length[s])) @ U.T
            v = v * self.em[em[s]]
            c[s] = v @ sq
            at[s + 1] = v / c[s]
        logl = np.log(c).sum()
        # backward with accumulation in the eigenbasis, one accumulator per (missing-data class, record limit)
        acc = {}
        bt = sq.copy()                               # beta~ = sqrt(pi) beta, beta_end = 1
        for s in range(nrow - 1, -1, -1):
            d, U, inv_dd = self.eig[kill_cls[s]]
            D = length[s]
            ex = np.exp(d * D)
            ah = at[s] @ U
            bh = U.T @ (self.em[em[s]] * bt)
            if D > 0:
                Phi = (ex[:, None] - ex[None, :]) * inv_dd
                Phi[np.diag_indices(K)] = D * ex
                key = (kill_cls[s], int(limit[s]))
                if key not in acc: acc[key] = np.zeros((K, K))
                acc[key] += (np.outer(ah, bh) * Phi) / c[s]
            bt = (U @ (bh * ex)) / c[s]
        gamma0 = at[0] * bt                          # posterior of the state at the start of the sequence
        gamma0 = gamma0 / gamma0.sum()
        return self._totals(acc, gamma0, logl)

    def _totals(self, acc, gamma0, logl):
        g = self.g
        K, E = g.K, g.E
        r2 = 2 * self.rho
        ind = (g.ep[:, None] == np.arange(E)[None, :]).astype(float)             # [K,E] cell in epoch e
        wE = ind * g.w[:, None]
        W = np.vstack([np.zeros((1, E)), np.cumsum(wE, 0)])                       # W[k] = sum_{k' < k} |cell k' n e|
        dbar = g.d / g.la[:, None]
        rbar = g.r / g.la[:, None]
        cum = lambda x: np.vstack([np.zeros((1, x.shape[1])), np.cumsum(x, 0)])[:-1]
        within = (2 * g.sw + g.uw)[:, None]
        inv_opp = r2 * (cum(g.d) + (2 * g.sw[:, None] * dbar + g.uw[:, None] * (dbar + wE / 3.0)) / g.pi[:, None])
        inv_cc = r2 * (cum(g.la[:, None] * ind) + within * ind / g.pi[:, None])
        inv_rc = r2 * (cum(g.r) + within * rbar / g.pi[:, None])
        Le_s = np.stack([g.Le(g.sbar, e) for e in range(E)], 1)                    # |[0, sbar] n e| (exact for cells outside e)
        out = {k: np.zeros(E) for k in ("coal_count", "coal_opp", "rec_count", "rec_opp")}
        occupancy = np.zeros(K)
        for (cls, lim), C in acc.items():
            d, U, _ = self.eig[cls]
            M = U @ C @ U.T                           # int alpha~_x(a) beta~_x(b) dx
            O = np.diag(M).copy()
            N = self.S * M                            # expected jumps a -> b
            Nd_to = np.tril(N, -1).sum(0)             # a > b
            Nu_to = np.triu(N, 1).sum(0)              # a < b
            Nu_from = np.triu(N, 1).sum(1)
            rec_mask = (np.arange(E) <= lim).astype(float)
            occupancy += O
            opp = O @ inv_opp + Nd_to @ dbar + Nu_from @ (dbar + ind * (g.lat / g.la)[:, None] - W[1:]) \
                + Nu_to @ (W[:-1] + ind * (g.pt / g.pi)[:, None])
            cc = O @ inv_cc + (Nd_to + Nu_to) @ ind
            rc = O @ inv_rc + Nd_to @ rbar + Nu_from @ rbar
            ro = O @ (2 * Le_s)
            out["coal_opp"] += opp * rec_mask
            out["coal_count"] += cc * rec_mask
            out["rec_count"] += rc * rec_mask
            out["rec_opp"] += ro * rec_mask          # exact only while the limit does not vary along a stretch
        out["coal_opp"] += gamma0 @ Le_s
        out["coal_count"] += gamma0 @ ind
        out["logl"] = float(logl)
        out["occupancy"] = occupancy
        out["gamma0"] = gamma0
        return out


def estimates(res, ne0=None, rho0=None):
    """the M-step ratios the .out file holds (src/pfparam.cpp:500-527), with the reference's pseudo-counts when the
    starting values are given (src/count.cpp:161-227: coal count 1/(2Ne), opportunity 1; recombination count rho, opportunity 1)"""
    cc, co = res["coal_count"].copy(), res["coal_opp"].copy()
    rc, ro = res["rec_count"].sum(), res["rec_opp"].sum()
    if ne0 is not None:
        cc = cc + 1.0 / (2 * np.asarray(ne0, float)); co = co + 1.0
    if rho0 is not None:
        rc += rho0; ro += 1.0
    return co / (2 * cc), rc / ro
