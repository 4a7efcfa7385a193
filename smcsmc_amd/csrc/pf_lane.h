// smcsmc_amd/csrc/pf_lane.h -- the per-lane LDS tree as the kernels use it: LDS carve-up, model tables,
// the SMC' genealogy update, tracked branch length and site likelihood on the LDS tree.
// Depends on PF_BS (the workgroup size = LDS stride of the per-lane columns), like pf_device.h.
#pragma once
#include "pf_device.h"
#include "pf_types.h"

using namespace pf;

// delayed importance weights: adjustWeightsWithDelay / applyDelayedAdjustment (particle.hpp:185-209)
struct DStore {
    double* pos; double* fac; double* delta; int* k;   // column of this particle: element i at [i * Np]
    long long Np;
    int count;
    double total;
    int cap, evict;        // KArgs::dcap, KArgs::delay_evict
    Ctrl* ctrl;            // error flag and the two statistics of the store
};
template <class KA>
__device__ __forceinline__ void d_bind(DStore& d, const KA& A, const DState& st, long long p) {
    d.pos = st.dpos + p; d.fac = st.dfac + p; d.delta = st.ddelta + p; d.k = st.dk + p; d.Np = A.Np;
    d.cap = A.dcap; d.evict = A.delay_evict; d.ctrl = A.ctrl;
}
__device__ __forceinline__ void d_apply_earliest(DStore& d, double& w_pilot) {
    int m = 0;
    double pm = d.pos[0];
    for (int i = 1; i < d.count; ++i) { double pi = d.pos[(size_t)i * d.Np]; if (pi < pm) { pm = pi; m = i; } }
    double f = d.fac[(size_t)m * d.Np];
    w_pilot *= f;
    d.total /= f;
    int km = d.k[(size_t)m * d.Np];
    if (km > 1) {
        double dl = d.delta[(size_t)m * d.Np];
        d.pos[(size_t)m * d.Np] = pm + 2 * dl;
        d.delta[(size_t)m * d.Np] = 2 * dl;
        d.k[(size_t)m * d.Np] = km - 1;
    } else {
        int last = --d.count;
        d.pos[(size_t)m * d.Np] = d.pos[(size_t)last * d.Np];
        d.fac[(size_t)m * d.Np] = d.fac[(size_t)last * d.Np];
        d.delta[(size_t)m * d.Np] = d.delta[(size_t)last * d.Np];
        d.k[(size_t)m * d.Np] = d.k[(size_t)last * d.Np];
    }
}
__device__ __forceinline__ void d_adjust_with_delay(DStore& d, double& w_post, double& w_pilot, double adj, double delay, double cur) {
    w_post *= adj;
    if ((adj > 0.99 && adj < 1.01) || (delay <= 1)) { w_pilot *= adj; return; }
    if (d.count == d.cap) {
        // a full store stops the run like every other bounded ring (the reference's heap is unbounded); with delay_evict the
        // earliest factor is applied ahead of its position instead (an entry leaves only with its third part), and counted
        if (!d.evict) { if (!d.ctrl->err) d.ctrl->err = ERR_DELAY_OVERFLOW; }
        while (d.count == d.cap) {
            d_apply_earliest(d, w_pilot);
            if (d.evict) atomicAdd(&d.ctrl->n_delay_evict, 1ull);
        }
    }
    d.total *= adj;
    double final_pos = cur + delay;
    double delta = (final_pos - cur) / 7.0;
    int i = d.count++;
    d.pos[(size_t)i * d.Np] = cur + delta;
    d.fac[(size_t)i * d.Np] = dexp(dlog(adj) * (1.0 / 3));
    d.delta[(size_t)i * d.Np] = delta;
    d.k[(size_t)i * d.Np] = 3;
    if (d.count > d.ctrl->delay_peak) atomicMax(&d.ctrl->delay_peak, d.count);
}

// LDS carve-up shared by k_init / k_extend
struct Smem {
    double* S; double* t0; double* t1; double* T; double* I; int* RF; int8_t* C;
};
__device__ __forceinline__ Smem carve(double* base, int n, int E) {
    Smem m;
    m.S = base;
    m.t0 = m.S + (size_t)(n - 1) * PF_BS;
    m.t1 = m.t0 + (size_t)(n - 1) * PF_BS;
    m.T = m.t1 + (size_t)(n - 1) * PF_BS;
    m.I = m.T + E;
    m.RF = (int*)(m.I + E);
    m.C = (int8_t*)(m.RF + E + (E & 1));
    return m;
}
__host__ __device__ static size_t smem_bytes(int n, int E) {
    return (size_t)3 * (n - 1) * PF_BS * 8 + (size_t)2 * E * 8 + (size_t)(E + (E & 1)) * 4 + (size_t)2 * (n - 1) * PF_BS;
}

__device__ __forceinline__ void load_model(const KArgs& A, Smem& m) {
    for (int e = threadIdx.x; e < A.E; e += blockDim.x) {
        m.T[e] = A.T[e];
        m.I[e] = A.inv2N[e];
        m.RF[e] = A.recflags[e];
    }
}

__device__ __forceinline__ Lane make_lane(const KArgs& A, Smem& m, long long p) {
    Lane ln;
    ln.S = m.S + threadIdx.x;
    ln.C = m.C + threadIdx.x;
    ln.T = m.T; ln.I = m.I; ln.RF = m.RF; ln.H = A.Hc;
    ln.E = A.E; ln.n = A.n;
    ln.L = A.L; ln.mu = A.mu; ln.rho = A.rho;
    ln.seed = A.seed;
    ln.slot = (unsigned)p;
    ln.stream = 0;
    ln.ctr = 0; ln.ebuf = 0; ln.Ltree = 0;
    ln.vbc = A.vb_coal; ln.upd_fac = 1.0;
    return ln;
}

// One genealogy update (SMC'): sample the recombination point, coalesce the floating lineage
// against the old tree, re-attach.  Mirrors oracle Filter::genealogy_update step by step.
__device__ __forceinline__ void sample_point(Lane& ln, int* rp_out, int* sb_out, double* h_out) {
    const int n = ln.n;
    double r = uni(ln) * ln.Ltree;
    double prev = 0.0, h = 0.0;
    int lin = 0;
    for (int ri = 0; ri < n - 1; ++ri) {
        int k = n - ri;
        double sr = LS(ln, ri);
        double d = sr - prev;
        double seg = (double)k * d;
        if (r < seg || ri == n - 2) {
            double q = r / d;
            lin = min((int)q, k - 1);
            h = prev + (q - (double)lin) * d;
            if (!(h < sr)) h = prev;
            break;
        }
        r -= seg;
        prev = sr;
    }
    lineages_at(ln, n - 1, h, lin, rp_out, sb_out);
    *h_out = h;
}

// samplePoint with height-band weights (particle.cpp:1020-1126) on the LDS tree; pieces = (time slice) x (band),
// ascending in height.  Same operation order as r_sample_point_biased and the oracle's biased branch; `iw_out` = target
// density over sampled density.  bH = 0, h1..hk, +inf; bS = strengths; nb = number of bands.
__device__ __forceinline__ void sample_point_biased(Lane& ln, const double* bH, const double* bS, int nb, int* rp_out, int* sb_out,
                                                    double* h_out, double* iw_out) {
    const int n = ln.n;
    double Lw = 0.0;
    {
        double pv = 0.0;
        int b = 0;
        for (int ri = 0; ri < n - 1; ++ri) {
            int k = n - ri;
            double top = LS(ln, ri);
            while (b + 1 < nb && bH[b + 1] <= pv) ++b;
            int bb = b;
            for (;;) {
                double lo_ = pv > bH[bb] ? pv : bH[bb];
                double hi_ = top < bH[bb + 1] ? top : bH[bb + 1];
                if (hi_ > lo_) Lw += ((double)k * bS[bb]) * (hi_ - lo_);
                if (bH[bb + 1] >= top || bb + 1 >= nb) break;
                ++bb;
            }
            pv = top;
        }
    }
    double r = uni(ln) * Lw;
    double l_lo = 0, l_hi = 0, l_str = 1;
    int l_k = 1;
    bool sel = false;
    double pv = 0.0;
    int b = 0;
    for (int ri = 0; ri < n - 1 && !sel; ++ri) {
        int k = n - ri;
        double top = LS(ln, ri);
        while (b + 1 < nb && bH[b + 1] <= pv) ++b;
        int bb = b;
        for (;;) {
            double lo_ = pv > bH[bb] ? pv : bH[bb];
            double hi_ = top < bH[bb + 1] ? top : bH[bb + 1];
            if (hi_ > lo_) {
                double wlen = ((double)k * bS[bb]) * (hi_ - lo_);
                l_lo = lo_; l_hi = hi_; l_str = bS[bb]; l_k = k;
                if (r < wlen) { sel = true; break; }
                r -= wlen;
            }
            if (bH[bb + 1] >= top || bb + 1 >= nb) break;
            ++bb;
        }
        pv = top;
    }
    double q = r / (l_str * (l_hi - l_lo));
    int lin = min((int)q, l_k - 1);
    if (lin < 0) lin = 0;
    double h = l_lo + (q - (double)lin) * (l_hi - l_lo);
    if (!(h < l_hi)) h = l_lo;
    double sampled = l_str / Lw;
    double target = 1.0 / ln.Ltree;
    *iw_out = target / sampled;
    lineages_at(ln, n - 1, h, lin, rp_out, sb_out);
    *h_out = h;
}

// samplePoint with a recombination guide (particle.cpp:942-1126) on the LDS tree: every branch carries a relative rate --
// leaf: the guide's rate `lr[i]` for that sample in the particle's segment; binary node: the mean of its children; the two
// branches below the root: the larger of the two -- times the strength of its height band.  Pieces are visited branch by
// branch in slot order, bands ascending; same operation order as r_sample_point_guided and the oracle.  `tmp` = per-lane
// LDS column of n-1 doubles (node rates).  rho_ratio = true rate / guide rate of the segment.
__device__ __forceinline__ void sample_point_guided(Lane& ln, const double* bH, const double* bS, int nb, const double* lr,
                                                    double rho_ratio, int* rp_out, int* sb_out, double* h_out, double* iw_out,
                                                    double* rbiw_out, double* tmp) {
    const int n = ln.n;
    auto rate_of = [&](int id) __attribute__((always_inline)) { return id < n ? lr[id] : tmp[(id - n) * PF_BS]; };
    for (int rr = 0; rr < n - 1; ++rr) tmp[rr * PF_BS] = (rate_of(LC(ln, rr, 0)) + rate_of(LC(ln, rr, 1))) * 0.5;
    const double ra = rate_of(LC(ln, n - 2, 0)), rbb = rate_of(LC(ln, n - 2, 1));
    const double rroot = ra > rbb ? ra : rbb;
    double Lw = 0.0;
    for (int rr = 0; rr < n - 1; ++rr) {
        const double hi_b = LS(ln, rr);
        for (int sdx = 0; sdx < 2; ++sdx) {
            const int c = LC(ln, rr, sdx);
            const double rb = (rr == n - 2) ? rroot : rate_of(c);
            const double lo_b = node_h(ln, c);
            for (int b = 0; b < nb; ++b) {
                double lo_ = lo_b > bH[b] ? lo_b : bH[b];
                double hi_ = hi_b < bH[b + 1] ? hi_b : bH[b + 1];
                if (hi_ > lo_) Lw += (rb * bS[b]) * (hi_ - lo_);
                if (bH[b + 1] >= hi_b) break;
            }
        }
    }
    double rr_ = uni(ln) * Lw;
    double l_lo = 0, l_hi = 0, l_wt = 1;
    bool sel = false;
    int g_rp = 0, g_sb = 0;
    for (int rr = 0; rr < n - 1 && !sel; ++rr) {
        const double hi_b = LS(ln, rr);
        for (int sdx = 0; sdx < 2 && !sel; ++sdx) {
            const int c = LC(ln, rr, sdx);
            const double rb = (rr == n - 2) ? rroot : rate_of(c);
            const double lo_b = node_h(ln, c);
            for (int b = 0; b < nb; ++b) {
                double lo_ = lo_b > bH[b] ? lo_b : bH[b];
                double hi_ = hi_b < bH[b + 1] ? hi_b : bH[b + 1];
                if (hi_ > lo_) {
                    double wt = rb * bS[b];
                    double wlen = wt * (hi_ - lo_);
                    l_lo = lo_; l_hi = hi_; l_wt = wt; g_rp = rr; g_sb = sdx;
                    if (rr_ < wlen) { sel = true; break; }
                    rr_ -= wlen;
                }
                if (bH[b + 1] >= hi_b) break;
            }
        }
    }
    double h = l_lo + rr_ / l_wt;
    if (!(h < l_hi)) h = l_lo;
    if (h < l_lo) h = l_lo;
    const double sampled = l_wt / Lw;
    const double target = 1.0 / ln.Ltree;
    double iw = target / sampled;
    iw *= rho_ratio;                                  // the position was drawn at the guide's rate
    double rbiw = 1.0;
    if (nb > 1) {
        // importance weight of the height bias alone (particle.cpp:1113-1121)
        double Lrw = 0.0, pv = 0.0;
        for (int ri = 0; ri < n - 1; ++ri) {
            const int k = n - ri;
            const double top = LS(ln, ri);
            for (int b = 0; b < nb; ++b) {
                double lo_ = pv > bH[b] ? pv : bH[b];
                double hi_ = top < bH[b + 1] ? top : bH[b + 1];
                if (hi_ > lo_) Lrw += ((double)k * bS[b]) * (hi_ - lo_);
                if (bH[b + 1] >= top) break;
            }
            pv = top;
        }
        int idx = 0;
        while (idx + 1 < nb && bH[idx + 1] < h) ++idx;
        const double recomb_density = bS[idx] / Lrw;
        rbiw = target / recomb_density;
    }
    *iw_out = iw; *rbiw_out = rbiw;
    *rp_out = g_rp; *sb_out = g_sb;
    *h_out = h;
}

// sample_next_base under a guide: the rate of the particle's segment, the draw limited to the segment
// (particle.cpp:1203-1232); without a guide (gK == 0) the plain draw
__device__ __forceinline__ double sample_next_base_guided(Lane& ln, double x, int gK, const double* gpos, const double* grho, int ridx) {
    if (gK == 0) return sample_next_base(ln, x);
    const double rho0 = ln.rho, L0 = ln.L;
    ln.rho = grho[ridx];
    if (ridx + 1 < gK) { const double nxt = gpos[ridx + 1]; if (nxt < L0) ln.L = nxt; }
    const double nb = sample_next_base(ln, x);
    ln.rho = rho0; ln.L = L0;
    return nb;
}

// get_descendants (descendants.hpp:22-33): samples below node `id`; `tmp` = per-lane LDS scratch of n-1 doubles
__device__ __forceinline__ unsigned lane_desc_mask(const Lane& ln, int id, double* tmp) {
    const int n = ln.n;
    if (id < n) return 1u << id;
    unsigned res = 0;
    for (int r = 0; r <= id - n; ++r) {
        int c0 = LC(ln, r, 0), c1 = LC(ln, r, 1);
        unsigned m0 = c0 < n ? (1u << c0) : (unsigned)__double_as_longlong(tmp[(c0 - n) * PF_BS]);
        unsigned m1 = c1 < n ? (1u << c1) : (unsigned)__double_as_longlong(tmp[(c1 - n) * PF_BS]);
        res = m0 | m1;
        tmp[r * PF_BS] = __longlong_as_double((long long)res);
    }
    return res;
}

// With bH != nullptr the cut point is drawn with height-band weights (focused sampling) and *iw_out receives the
// importance weight of the draw.
__device__ __forceinline__ void genealogy_update(Lane& ln, double* h_out, double* tc_out, double* sp_out, bool* changed_out,
                                                 unsigned* desc_out = nullptr, double* tmp = nullptr,
                                                 const double* bH = nullptr, const double* bS = nullptr, int nb = 1,
                                                 double* iw_out = nullptr, const double* lr = nullptr, double rho_ratio = 1.0,
                                                 double* rbiw_out = nullptr, unsigned* desc_new_out = nullptr) {
    const int n = ln.n;
    int rp = 0, sb = 0;
    double h;
    prefetch_update_uniforms(ln);       // the caller drops what is left after the sample_next_base that follows
    if (lr) sample_point_guided(ln, bH, bS, nb, lr, rho_ratio, &rp, &sb, &h, iw_out, rbiw_out, tmp);
    else if (bH) { sample_point_biased(ln, bH, bS, nb, &rp, &sb, &h, iw_out); if (rbiw_out) *rbiw_out = *iw_out; }
    else sample_point(ln, &rp, &sb, &h);
    if (desc_out) *desc_out = lane_desc_mask(ln, LC(ln, rp, sb), tmp);
    *h_out = h;
    double tc = coalesce_up(ln, [&](int k) { return LS(ln, k); }, n - 1, n, h);
    *tc_out = tc;
    double Sp = LS(ln, rp);
    int b_id = LC(ln, rp, sb), s_id = LC(ln, rp, 1 - sb);
    bool p_was_root = (rp == n - 2);
    remove_rank(ln, n - 1, rp, s_id, &b_id, &s_id);
    int ni = n - 2;
    int troot = p_was_root ? s_id : n + (ni - 1);
    int pr = -1, ps = 0;
    int nslots = lineages_at(ln, ni, tc, -1, &pr, &ps);
    bool has_root = tc >= node_h(ln, troot);
    bool has_stub = tc < Sp;
    int k = nslots + (has_root ? 1 : 0) + (has_stub ? 1 : 0);
    double u = uni(ln);
    int idx = min((int)(u * (double)k), k - 1);
    *sp_out = Sp;
    *changed_out = !(has_stub && idx == k - 1);
    double h_ins = tc;
    int pr_ins = -1, ps_ins = 0;
    if (idx < nslots) {
        lineages_at(ln, ni, tc, idx, &pr_ins, &ps_ins);
    } else if (!(has_root && idx == nslots)) {
        h_ins = Sp;
        if (!p_was_root) {
            int R = 0;
            while (R < ni && LS(ln, R) <= Sp) ++R;
            bool found = false;
            for (int rr = R; rr < ni && !found; ++rr)
                for (int s = 0; s < 2 && !found; ++s) {
                    int id = LC(ln, rr, s);
                    if ((id < n || id - n < R) && id == s_id) { found = true; pr_ins = rr; ps_ins = s; }
                }
        }
    }
    if (desc_new_out && desc_out) {
        // samples below the node this update creates (masks taken on the pruned tree, before the insertion)
        unsigned dn = *desc_out;
        if (idx < nslots) dn |= lane_desc_mask(ln, LC(ln, pr_ins, ps_ins), tmp);
        else if (has_root && idx == nslots) dn = (1u << n) - 1u;
        *desc_new_out = dn;
    }
    insert_node(ln, ni, h_ins, b_id, pr_ins, ps_ins, troot);
    ln.Ltree = tree_length(ln, n);
}

__device__ __forceinline__ double tracked_len_lane(const Lane& ln, const int8_t* data, double* tmp) {
    // particle.cpp:699-730; tmp holds the per-internal-node values (stride PF_BS)
    const int n = ln.n;
    double total = 0.0;
    for (int r = 0; r < n - 1; ++r) {
        int c0 = LC(ln, r, 0), c1 = LC(ln, r, 1);
        double sr = LS(ln, r);
        double l = c0 < n ? (data[c0] >= 0 ? 0.0 : -1.0) : tmp[(c0 - n) * PF_BS];
        double rr = c1 < n ? (data[c1] >= 0 ? 0.0 : -1.0) : tmp[(c1 - n) * PF_BS];
        if (l >= 0.0) l += sr - node_h(ln, c0);
        if (rr >= 0.0) rr += sr - node_h(ln, c1);
        double v;
        if (l >= 0.0 && rr >= 0.0) { total = l + rr; v = total; }
        else if (l >= 0.0) v = l;
        else v = rr;
        tmp[r * PF_BS] = v;
    }
    return total;
}

__device__ __forceinline__ double site_lik_lane(const Lane& ln, unsigned one_mask, unsigned zero_mask, bool anc,
                                                double* t0, double* t1) {
    // particle.cpp:625-680.  Leaf i: L0 = (state==1 ? 0 : 1), L1 = (state==0 ? 0 : 1);
    // one_mask bit i <=> state==1, zero_mask bit i <=> state==0 (missing: neither).
    const int n = ln.n;
    for (int r = 0; r < n - 1; ++r) {
        int c0 = LC(ln, r, 0), c1 = LC(ln, r, 1);
        double sr = LS(ln, r);
        double tl = sr - node_h(ln, c0);
        double trr = sr - node_h(ln, c1);
        double pl = fastexp(-tl * ln.mu);
        double pr = fastexp(-trr * ln.mu);
        double a0, a1, b0, b1;
        if (c0 < n) { a0 = (one_mask >> c0) & 1 ? 0.0 : 1.0; a1 = (zero_mask >> c0) & 1 ? 0.0 : 1.0; }
        else { a0 = t0[(c0 - n) * PF_BS]; a1 = t1[(c0 - n) * PF_BS]; }
        if (c1 < n) { b0 = (one_mask >> c1) & 1 ? 0.0 : 1.0; b1 = (zero_mask >> c1) & 1 ? 0.0 : 1.0; }
        else { b0 = t0[(c1 - n) * PF_BS]; b1 = t1[(c1 - n) * PF_BS]; }
        t0[r * PF_BS] = (a0 * pl + a1 * (1 - pl)) * (b0 * pr + b1 * (1 - pr));
        t1[r * PF_BS] = (a1 * pl + a0 * (1 - pl)) * (b1 * pr + b0 * (1 - pr));
    }
    double p0 = anc ? 1.0 : 0.5, p1 = anc ? 0.0 : 0.5;
    return t0[(n - 2) * PF_BS] * p0 + t1[(n - 2) * PF_BS] * p1;
}

