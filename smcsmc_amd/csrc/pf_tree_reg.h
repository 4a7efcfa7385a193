// smcsmc_amd/csrc/pf_tree_reg.h -- register-resident local tree for small sample sizes.
//
// Same algorithms, same IEEE-754 operations in the same order as the LDS-backed versions in
// pf_device.h (and therefore the same bits as the oracle); only the storage differs: the n-1 node
// heights and 2(n-1) child ids live in VGPRs, every loop over ranks is fully unrolled with a
// compile-time index, and the few genuinely run-time indices become v_cndmask select chains.
// This removes the dependent LDS round trips (~64+ cycles each, one wavefront per SIMD, nothing to
// hide them behind) that dominated k_extend.
#pragma once
#include "pf_device.h"

namespace pf {

template <int NM>   // NM = maximum number of haplotypes handled by this instantiation
struct RTree {
    static constexpr int NI = NM - 1;
    double S[NI];
    int C0[NI], C1[NI];

    // Bit-mask select: written as `(r == k) ? S[k] : v` the compiler folds the chain back into a dynamically indexed
    // load, which puts the whole tree into scratch memory and a scratch round trip into the coalescence walk.
    __device__ __forceinline__ double getS(int r) const {
        long long bits = __double_as_longlong(S[0]);
#pragma unroll
        for (int k = 1; k < NI; ++k) {
            const long long m = -(long long)(r == k);
            bits = (bits & ~m) | (__double_as_longlong(S[k]) & m);
        }
        return __longlong_as_double(bits);
    }
    __device__ __forceinline__ void setS(int r, double x) {
        const long long xb = __double_as_longlong(x);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const long long m = -(long long)(r == k);
            S[k] = __longlong_as_double((__double_as_longlong(S[k]) & ~m) | (xb & m));
        }
    }
    __device__ __forceinline__ int getC(int r, int s) const {
        const int ms = -(int)(s != 0);
        int v = 0;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int m = -(int)(r == k);
            v |= ((C0[k] & ~ms) | (C1[k] & ms)) & m;
        }
        return v;
    }
    __device__ __forceinline__ void setC(int r, int s, int x) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int m0 = -(int)(r == k && s == 0), m1 = -(int)(r == k && s == 1);
            C0[k] = (C0[k] & ~m0) | (x & m0);
            C1[k] = (C1[k] & ~m1) | (x & m1);
        }
    }
};

// per-lane context that is not the tree (epoch tables stay in LDS: they are indexed by epoch)
struct RCtx {
    const double* T;
    const double* I;
    const double* H;          // cumulative coalescence intensity at the epoch starts (LDS)
    double u_nb;              // fourth uniform of the update in progress (next recombination position)
    int E, n;
    double L, mu, rho;
    unsigned long long seed;
    unsigned slot, stream;
    unsigned long long ctr;
    double ebuf;
    double Ltree;
    // focused sampling (particle.cpp:1020-1126): nb bands, band b = [H[b], H[b+1]) weighs S[b]
    int nb;
    const double* bH;
    const double* bS;
    double last_iw;
    double last_rbiw;     // importance weight of the height bias alone (recombination_bias_importance_weight_)
    // recombination guide (RecombinationBias, pfparam.hpp:96-223); gK == 0: none
    int gK, ridx;
    const double* gpos;   // [gK] segment starts
    const double* grho;   // [gK] sampling rate per segment
    const double* gleaf;  // [gK*n] relative rate per sample
    int g_rp, g_sb;       // slot chosen by the guided sampler
    const double* vbc;    // variational-Bayes factor per epoch of a coalescence (particle.cpp:266-272), or null
    double upd_fac;       // the factor of the current update
    unsigned last_desc;   // samples below the branch cut by the last update (only computed when want_desc)
    unsigned last_desc_new;   // samples below the node created by the last update (only computed when want_desc)
    bool want_desc;
    bool want_desc_new;   // -arg only
    // draw table of the row pipeline (draw_role, pf_hip.hip): the blocks of this slot's stream below `tab_end` are in `tab`
    // (entry c % PF_DRAW_RING = first uniform of block c and minus the logarithm of its second one).  The four numbers of
    // the next update are requested one update ahead (r_draws_prefetch) and wait here.
    const double2* tab;       // null: no table
    unsigned tab_end;         // low word of the first block index that is not in the table
    unsigned pf_ctr;          // low word of the block index the prefetched numbers belong to
    bool pf_ok;
    bool draws_log;           // the second and fourth number of the update in progress are already -log(uniform)
    double pf_u0, pf_e1, pf_u2, pf_e3;
    // bucket tables of the two searches of an update (r_search_lut): 256 bytes each in LDS, or null
    const unsigned char* lutT; const unsigned char* lutH;
    int kbT, kbH;
};

#define PF_DRAW_RING 32        // blocks per slot kept in the draw table (sixteen genealogy updates)

// request the table entries of the update that starts at block cx.ctr (a round trip to the L2 or beyond: issued an update
// ahead, or in the prologue of the row); without an entry for both blocks the update computes its own
__device__ __forceinline__ void r_draws_prefetch(RCtx& cx) {
    cx.pf_ok = false;
    if (cx.tab != nullptr && (int)(cx.tab_end - (unsigned)cx.ctr) >= 2) {
        const double2 a = cx.tab[(unsigned)cx.ctr & (PF_DRAW_RING - 1)];
        const double2 b = cx.tab[((unsigned)cx.ctr + 1u) & (PF_DRAW_RING - 1)];
        cx.pf_u0 = a.x; cx.pf_e1 = a.y; cx.pf_u2 = b.x; cx.pf_e3 = b.y;
        cx.pf_ctr = (unsigned)cx.ctr;
        cx.pf_ok = true;
    }
}

__device__ __forceinline__ double r_uni(RCtx& cx) { return philox_uniform(cx.seed, cx.slot, cx.stream, cx.ctr++); }
// Largest e with tab[e] <= t, for an ascending table of PF_EPAD = 64 doubles in LDS, padded with +inf behind its E
// entries (tab[0] <= t).  Three levels of a four-way search; since the table ascends, the new lower bound is
// lo + stride * #(probes <= t).  No bounds checks (the padding takes their place) and the three probes of a level
// share one address register: a search is about thirty instructions, and with one wavefront per SIMD the
// instruction count is the cost.
#define PF_EPAD 64
__device__ __forceinline__ int r_search4(const double* tab, double t) {
    int lo = 0;
#pragma unroll
    for (int stride = 16; stride >= 1; stride >>= 2) {
        const double* q = tab + lo;
        const double v1 = q[stride], v2 = q[2 * stride], v3 = q[3 * stride];
        lo += stride * ((v1 <= t ? 1 : 0) + (v2 <= t ? 1 : 0) + (v3 <= t ? 1 : 0));
    }
    return lo;
}
// N searches at once, level by level: the 3 N probes of a level are all in flight before the first is used
template <int N>
__device__ __forceinline__ void r_search4_batch(const double* tab, const double (&tv)[N], int (&lo)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) lo[k] = 0;
#pragma unroll
    for (int stride = 16; stride >= 1; stride >>= 2) {
        double v1[N], v2[N], v3[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double* q = tab + lo[k];
            v1[k] = q[stride]; v2[k] = q[2 * stride]; v3[k] = q[3 * stride];
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
            lo[k] += stride * ((v1[k] <= tv[k] ? 1 : 0) + (v2[k] <= tv[k] ? 1 : 0) + (v3[k] <= tv[k] ? 1 : 0));
    }
}
// The same answer from a bucket table: the upper sixteen bits of a non-negative double (exponent and four bits of the
// mantissa: sixteen buckets per octave) grow with the number, so `lut[key - kbase]` = the answer at the bucket's lower edge is
// a lower bound, and the host only supplies the table when no bucket holds more than two table entries (lut_build, pf_hip.hip):
// two probes finish the search.  One byte and two doubles from LDS in two round trips and a dozen instructions, where the
// four-way search takes nine doubles in three round trips and thirty instructions.
#define PF_LUT_N 256
__device__ __forceinline__ int r_search_lut(const double* tab, const unsigned char* lut, int kbase, double t) {
    int key = (int)(((unsigned)__double2hiint(t) >> 16) & 0x7fffu) - kbase;
    key = key < 0 ? 0 : (key > PF_LUT_N - 1 ? PF_LUT_N - 1 : key);
    const int e = lut[key];
    const double v1 = tab[e + 1], v2 = tab[e + 2];
    return e + (v1 <= t ? 1 : 0) + (v2 <= t ? 1 : 0);
}
template <int N>
__device__ __forceinline__ void r_search_lut_batch(const double* tab, const unsigned char* lut, int kbase, const double (&tv)[N], int (&lo)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int key = (int)(((unsigned)__double2hiint(tv[k]) >> 16) & 0x7fffu) - kbase;
        key = key < 0 ? 0 : (key > PF_LUT_N - 1 ? PF_LUT_N - 1 : key);
        lo[k] = lut[key];
    }
    double v1[N], v2[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { v1[k] = tab[lo[k] + 1]; v2[k] = tab[lo[k] + 2]; }
#pragma unroll
    for (int k = 0; k < N; ++k) lo[k] += (v1[k] <= tv[k] ? 1 : 0) + (v2[k] <= tv[k] ? 1 : 0);
}
// epoch containing time t: the largest e with T[e] <= t (T[0] = 0)
__device__ __forceinline__ int r_epoch_of(const RCtx& cx, double t) { return r_search4(cx.T, t); }
__device__ __forceinline__ double r_epoch_end(const RCtx& cx, int e) { return e + 1 < cx.E ? cx.T[e + 1] : PF_INF; }

template <int NM>
__device__ __forceinline__ double r_node_h(const RTree<NM>& t, int n, int id) { return id < n ? 0.0 : t.getS(id - n); }

template <int NM>
__device__ __forceinline__ double r_tree_length(const RTree<NM>& t, int n) {
    double acc = 0.0, prev = 0.0;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r)
        if (r < n - 1) {
            double s = t.S[r];
            acc += (double)(n - r) * (s - prev);
            prev = s;
        }
    return acc;
}

template <int NM>
__device__ __forceinline__ int r_lineages_at(const RTree<NM>& t, int n, int ni, double time, int want, int* pr, int* ps) {
    int R = 0;
#pragma unroll
    for (int k = 0; k < RTree<NM>::NI; ++k) R += (k < ni && t.S[k] <= time) ? 1 : 0;   // S is sorted: a prefix count
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r) {
        if (r >= R && r < ni) {
            int id0 = t.C0[r];
            if (id0 < n || id0 - n < R) {
                if (cnt == want) { *pr = r; *ps = 0; }
                ++cnt;
            }
            int id1 = t.C1[r];
            if (id1 < n || id1 - n < R) {
                if (cnt == want) { *pr = r; *ps = 1; }
                ++cnt;
            }
        }
    }
    return cnt;
}

template <int NM>
__device__ __forceinline__ void r_remove_rank(RTree<NM>& t, int n, int ni, int rp, int sib, int* a, int* b) {
    const int pid = n + rp;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r)
        if (r > rp && r < ni) {
            if (t.C0[r] == pid) t.C0[r] = sib;
            if (t.C1[r] == pid) t.C1[r] = sib;
        }
#pragma unroll
    for (int r = 0; r + 1 < RTree<NM>::NI; ++r)
        if (r >= rp && r + 1 < ni) {
            t.S[r] = t.S[r + 1];
            t.C0[r] = t.C0[r + 1];
            t.C1[r] = t.C1[r + 1];
        }
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r)
        if (r < ni - 1) {
            if (t.C0[r] > pid) t.C0[r] -= 1;
            if (t.C1[r] > pid) t.C1[r] -= 1;
        }
    if (*a > pid) *a -= 1;
    if (*b > pid) *b -= 1;
}

template <int NM>
__device__ __forceinline__ void r_insert_node(RTree<NM>& t, int n, int ni, double h, int fl, int pr, int ps, int root_id) {
    int rn = 0;
#pragma unroll
    for (int k = 0; k < RTree<NM>::NI; ++k) rn += (k < ni && t.S[k] <= h) ? 1 : 0;
    const int nid = n + rn;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r)
        if (r < ni) {
            if (t.C0[r] >= nid) t.C0[r] += 1;
            if (t.C1[r] >= nid) t.C1[r] += 1;
        }
    if (fl >= nid) fl += 1;
    if (root_id >= nid) root_id += 1;
#pragma unroll
    for (int r = RTree<NM>::NI - 1; r >= 1; --r)
        if (r <= ni && r > rn) {
            t.S[r] = t.S[r - 1];
            t.C0[r] = t.C0[r - 1];
            t.C1[r] = t.C1[r - 1];
        }
    int target;
    if (pr >= 0) {
        if (pr >= rn) pr += 1;
        target = t.getC(pr, ps);
        t.setC(pr, ps, nid);
    } else {
        target = root_id;
    }
    t.setS(rn, h);
    t.setC(rn, 0, fl);
    t.setC(rn, 1, target);
}

template <int NM, bool TAB = false>
__device__ __forceinline__ double r_coalesce_up(RCtx& cx, const RTree<NM>& t, int ns, int nl, double h, double u_refresh) {
    // Cumulative-intensity form of the walk (see coalesce_up in pf_device.h: same arithmetic, operation for
    // operation): one comparison per node passed instead of one per epoch passed, then the inverse of the piecewise
    // linear Hc by a search.  No loop is left: the node scan is unrolled over the (at most NI) ranks.
    // The epoch searches of the cut height and of every node are independent chains of LDS reads: issued together
    // they cost one search's latency (inside a loop over the nodes passed each would wait for the previous one).
    constexpr int NI = RTree<NM>::NI;
    double Hn_h;
    double tv[NI + 1];
    int ev[NI + 1];
#pragma unroll
    for (int r = 0; r < NI; ++r) tv[r] = t.S[r];     // unused ranks hold 0: a harmless search, no branch
    tv[NI] = h;
    if (TAB && cx.lutT) r_search_lut_batch<NI + 1>(cx.T, cx.lutT, cx.kbT, tv, ev);
    else r_search4_batch<NI + 1>(cx.T, tv, ev);
    double Hn[NI];
    {
        double hh[NI + 1], tt[NI + 1], ii[NI + 1];
#pragma unroll
        for (int r = 0; r <= NI; ++r) { hh[r] = cx.H[ev[r]]; tt[r] = cx.T[ev[r]]; ii[r] = cx.I[ev[r]]; }
#pragma unroll
        for (int r = 0; r < NI; ++r) Hn[r] = hh[r] + (tv[r] - tt[r]) * ii[r];
        hh[0] = hh[NI]; tt[0] = tt[NI]; ii[0] = ii[NI];
        Hn_h = hh[0] + (h - tt[0]) * ii[0];
    }
    double Hc = Hn_h;
    double lower = h, kd = 1.0, sn = PF_INF;
    bool stopped = false;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r) {
        const double sr = t.S[r];
        if (r < ns && !stopped && sr > h) {          // nodes above the cut in rank order: r = number of nodes passed
            const double k = (double)(nl - r);
            const double need = (Hn[r] - Hc) * k;
            if (!(cx.ebuf > need)) { stopped = true; kd = k; sn = sr; }
            else { cx.ebuf -= need; Hc = Hn[r]; lower = sr; }
        }
    }
    const double C = Hc + cx.ebuf / kd;
    const int es = (TAB && cx.lutH) ? r_search_lut(cx.H, cx.lutH, cx.kbH, C) : r_search4(cx.H, C);
    double t1 = cx.T[es] + (C - cx.H[es]) / cx.I[es];
    if (t1 < lower) t1 = lower;
    if (t1 > sn) t1 = sn;
    cx.ebuf = (TAB && cx.draws_log) ? u_refresh : -dlog(u_refresh);
    if (cx.vbc) cx.upd_fac *= cx.vbc[es];
    return t1;
}

template <bool SPARE, bool TAB = false>
__device__ __forceinline__ double r_sample_next_base(RCtx& cx, double x) {
    // with a guide: the rate of the particle's current segment, the draw limited to the segment (particle.cpp:1203-1232)
    double rho_here = cx.rho, seg_end = cx.L;
    if (cx.gK > 0) {
        rho_here = cx.grho[cx.ridx];
        if (cx.ridx + 1 < cx.gK) { double nxt = cx.gpos[cx.ridx + 1]; if (nxt < cx.L) seg_end = nxt; }
    }
    double rate = rho_here * cx.Ltree;
    double limit = seg_end - x;
    double need = limit * rate;
    if (cx.ebuf > need) {
        cx.ebuf -= need;
        return seg_end;
    }
    double nb = x + cx.ebuf / rate;
    if (SPARE && TAB && cx.draws_log) cx.ebuf = cx.u_nb;
    else cx.ebuf = -dlog(SPARE ? cx.u_nb : r_uni(cx));
    if (nb == x) {
        nb = __longlong_as_double(__double_as_longlong(x) + 1);
        if (x == 0.0) nb = 4.9406564584124654e-324;
    }
    if (nb > seg_end) nb = seg_end;
    return nb;
}

// samplePoint with height-band weights (particle.cpp:1020-1126); pieces = (time slice) x (band), ascending in
// height; same operation order as the oracle's biased branch of Filter::genealogy_update.
template <int NM>
__device__ __forceinline__ void r_sample_point_biased(RCtx& cx, const RTree<NM>& t, double u_point, double* h_out, int* lin_out) {
    const int n = cx.n;
    const int nb = cx.nb;
    double Lw = 0.0;
    {
        double pv = 0.0;
        int b = 0;
#pragma unroll
        for (int ri = 0; ri < RTree<NM>::NI; ++ri)
            if (ri < n - 1) {
                int k = n - ri;
                double top = t.S[ri];
                while (b + 1 < nb && cx.bH[b + 1] <= pv) ++b;
                int bb = b;
                for (;;) {
                    double lo_ = pv > cx.bH[bb] ? pv : cx.bH[bb];
                    double hi_ = top < cx.bH[bb + 1] ? top : cx.bH[bb + 1];
                    if (hi_ > lo_) Lw += ((double)k * cx.bS[bb]) * (hi_ - lo_);
                    if (cx.bH[bb + 1] >= top || bb + 1 >= nb) break;
                    ++bb;
                }
                pv = top;
            }
    }
    double r = u_point * Lw;
    double l_lo = 0, l_hi = 0, l_str = 1;
    int l_k = 1;
    bool sel = false;
    double pv = 0.0;
    int b = 0;
#pragma unroll
    for (int ri = 0; ri < RTree<NM>::NI; ++ri)
        if (ri < n - 1 && !sel) {
            int k = n - ri;
            double top = t.S[ri];
            while (b + 1 < nb && cx.bH[b + 1] <= pv) ++b;
            int bb = b;
            for (;;) {
                double lo_ = pv > cx.bH[bb] ? pv : cx.bH[bb];
                double hi_ = top < cx.bH[bb + 1] ? top : cx.bH[bb + 1];
                if (hi_ > lo_) {
                    double wlen = ((double)k * cx.bS[bb]) * (hi_ - lo_);
                    l_lo = lo_; l_hi = hi_; l_str = cx.bS[bb]; l_k = k;
                    if (r < wlen) { sel = true; break; }
                    r -= wlen;
                }
                if (cx.bH[bb + 1] >= top || bb + 1 >= nb) break;
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
    double target = 1.0 / cx.Ltree;
    cx.last_iw = target / sampled;
    *h_out = h;
    *lin_out = lin;
}

// samplePoint with a recombination guide (particle.cpp:942-1126): every branch carries a relative rate -- leaf: the guide's
// rate for that sample in the particle's segment; binary node: the arithmetic mean of its children; the two branches
// below the root: the larger of the two -- times the strength of its height band.  Pieces are visited branch by branch in
// slot order (rank ascending, child 0 then 1), bands ascending; same operation order as the oracle.
template <int NM>
__device__ __forceinline__ void r_sample_point_guided(RCtx& cx, const RTree<NM>& t, bool height_bias, double u_point, double* h_out) {
    constexpr int NI = RTree<NM>::NI;
    const int n = cx.n;
    const int nb = cx.nb;
    const double* lr = cx.gleaf + (size_t)cx.ridx * n;
    double bl[NM], bn[NI];
#pragma unroll
    for (int i = 0; i < NM; ++i) bl[i] = i < n ? lr[i] : 0.0;
    auto rate_of = [&](int id, int upto) __attribute__((always_inline)) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < NM; ++i) v = (id == i) ? bl[i] : v;
#pragma unroll
        for (int k = 0; k < NI; ++k) if (k < upto) v = (id - n == k) ? bn[k] : v;
        return v;
    };
#pragma unroll
    for (int rr = 0; rr < NI; ++rr) {
        bn[rr] = 0.0;
        if (rr < n - 1) bn[rr] = (rate_of(t.C0[rr], rr) + rate_of(t.C1[rr], rr)) * 0.5;
    }
    const int c0r = t.getC(n - 2, 0), c1r = t.getC(n - 2, 1);
    const double ra = rate_of(c0r, NI), rbb = rate_of(c1r, NI);
    const double rroot = ra > rbb ? ra : rbb;
    // pass 1: weighted length
    double Lw = 0.0;
#pragma unroll
    for (int rr = 0; rr < NI; ++rr)
        if (rr < n - 1) {
            const double hi_b = t.S[rr];
#pragma unroll
            for (int sdx = 0; sdx < 2; ++sdx) {
                const int c = sdx ? t.C1[rr] : t.C0[rr];
                const double rb = (rr == n - 2) ? rroot : rate_of(c, NI);
                const double lo_b = r_node_h(t, n, c);
                for (int b = 0; b < nb; ++b) {
                    double lo_ = lo_b > cx.bH[b] ? lo_b : cx.bH[b];
                    double hi_ = hi_b < cx.bH[b + 1] ? hi_b : cx.bH[b + 1];
                    if (hi_ > lo_) Lw += (rb * cx.bS[b]) * (hi_ - lo_);
                    if (cx.bH[b + 1] >= hi_b) break;
                }
            }
        }
    // pass 2: the piece that holds U * Lw
    double rr_ = u_point * Lw;
    double l_lo = 0, l_hi = 0, l_wt = 1;
    bool sel = false;
    int g_rp = 0, g_sb = 0;
#pragma unroll
    for (int rr = 0; rr < NI; ++rr)
        if (rr < n - 1 && !sel) {
            const double hi_b = t.S[rr];
#pragma unroll
            for (int sdx = 0; sdx < 2; ++sdx)
                if (!sel) {
                    const int c = sdx ? t.C1[rr] : t.C0[rr];
                    const double rb = (rr == n - 2) ? rroot : rate_of(c, NI);
                    const double lo_b = r_node_h(t, n, c);
                    for (int b = 0; b < nb; ++b) {
                        double lo_ = lo_b > cx.bH[b] ? lo_b : cx.bH[b];
                        double hi_ = hi_b < cx.bH[b + 1] ? hi_b : cx.bH[b + 1];
                        if (hi_ > lo_) {
                            double wt = rb * cx.bS[b];
                            double wlen = wt * (hi_ - lo_);
                            l_lo = lo_; l_hi = hi_; l_wt = wt; g_rp = rr; g_sb = sdx;
                            if (rr_ < wlen) { sel = true; break; }
                            rr_ -= wlen;
                        }
                        if (cx.bH[b + 1] >= hi_b) break;
                    }
                }
        }
    double h = l_lo + rr_ / l_wt;
    if (!(h < l_hi)) h = l_lo;
    if (h < l_lo) h = l_lo;
    const double sampled = l_wt / Lw;
    const double target = 1.0 / cx.Ltree;
    cx.last_iw = target / sampled;
    cx.last_iw *= cx.rho / cx.grho[cx.ridx];       // the position was drawn at the guide's rate
    cx.last_rbiw = 1.0;
    if (height_bias) {
        // importance weight of the height bias alone (particle.cpp:1113-1121)
        double Lrw = 0.0, pv = 0.0;
#pragma unroll
        for (int ri = 0; ri < NI; ++ri)
            if (ri < n - 1) {
                const int k = n - ri;
                const double top = t.S[ri];
                for (int b = 0; b < nb; ++b) {
                    double lo_ = pv > cx.bH[b] ? pv : cx.bH[b];
                    double hi_ = top < cx.bH[b + 1] ? top : cx.bH[b + 1];
                    if (hi_ > lo_) Lrw += ((double)k * cx.bS[b]) * (hi_ - lo_);
                    if (cx.bH[b + 1] >= top) break;
                }
                pv = top;
            }
        int idx = 0;
        while (idx + 1 < nb && cx.bH[idx + 1] < h) ++idx;
        const double recomb_density = cx.bS[idx] / Lrw;
        cx.last_rbiw = target / recomb_density;
    }
    cx.g_rp = g_rp; cx.g_sb = g_sb;
    *h_out = h;
}

// One SMC' genealogy update; mirrors genealogy_update() in pf_hip.hip / Filter::genealogy_update in the oracle.
template <int NM, bool BIASED, bool TAB = false>
__device__ __forceinline__ void r_genealogy_update(RCtx& cx, RTree<NM>& t, double* h_out, double* tc_out,
                                                   double* sp_out = nullptr, bool* changed_out = nullptr) {
    const int n = cx.n;
    double h = 0.0;
    int lin = 0;
    bool guided_pt = false;
    // the update's four uniforms: two Philox blocks, drawn together (philox_pair)
    // TAB: from the draw table when the numbers of exactly this update were requested in time (then the second and the fourth
    // arrive as logarithms), and the request for the next update goes out at once
    double u_point, u_refresh, u_attach;
    bool from_tab = false;
    if constexpr (TAB) {
        from_tab = cx.pf_ok && cx.pf_ctr == (unsigned)cx.ctr;
        cx.draws_log = from_tab;
        u_point = cx.pf_u0; u_refresh = cx.pf_e1; u_attach = cx.pf_u2; cx.u_nb = cx.pf_e3;
    }
    if (!from_tab) {
        philox_pair(cx.seed, cx.slot, cx.stream, cx.ctr, u_point, u_refresh);
        philox_pair(cx.seed, cx.slot, cx.stream, cx.ctr + 1, u_attach, cx.u_nb);
    }
    cx.ctr += 2;
    if constexpr (TAB) r_draws_prefetch(cx);
    if (BIASED && cx.gK > 0 && cx.stream == 0) {
        r_sample_point_guided(cx, t, cx.nb > 1, u_point, &h);
        guided_pt = true;
    } else if (BIASED) {
        r_sample_point_biased(cx, t, u_point, &h, &lin);
        cx.last_rbiw = cx.last_iw;
    } else {
        // the slice is found first, the one division follows (inside the unrolled scan every rank would carry its own)
        double r = u_point * cx.Ltree;
        double prev = 0.0;
        double sel_r = 0.0, sel_d = 1.0, sel_prev = 0.0, sel_sr = 0.0;
        int sel_k = 1;
        bool done = false;
#pragma unroll
        for (int ri = 0; ri < RTree<NM>::NI; ++ri) {
            if (!done && ri < n - 1) {
                int k = n - ri;
                double sr = t.S[ri];
                double d = sr - prev;
                double seg = (double)k * d;
                if (r < seg || ri == n - 2) {
                    sel_r = r; sel_d = d; sel_prev = prev; sel_sr = sr; sel_k = k;
                    done = true;
                } else {
                    r -= seg;
                    prev = sr;
                }
            }
        }
        {
            double q = sel_r / sel_d;
            lin = min((int)q, sel_k - 1);
            h = sel_prev + (q - (double)lin) * sel_d;
            if (!(h < sel_sr)) h = sel_prev;
        }
    }
    int rp = 0, sb = 0;
    if (guided_pt) { rp = cx.g_rp; sb = cx.g_sb; }
    else r_lineages_at(t, n, n - 1, h, lin, &rp, &sb);
    *h_out = h;
    unsigned tmask[RTree<NM>::NI + NM];          // -arg only: samples below every node id of the tree before the cut
    if (TAB || cx.want_desc) {
        // (on the rows of k_sweep always: a dozen selects that the scheduler can place beside the search that follows, where
        //  a branch on the flag made them a block of their own in the chain; the local map is on by default)
        // get_descendants (descendants.hpp:22-33) of the cut branch on the tree before it changes: masks bottom-up
        unsigned below[RTree<NM>::NI];
        unsigned cut = 0;
#pragma unroll
        for (int r = 0; r < RTree<NM>::NI; ++r) {
            below[r] = 0;
            if (r < n - 1) {
                const int c0 = t.C0[r], c1 = t.C1[r];
                unsigned m0 = c0 < n ? (1u << c0) : 0u, m1 = c1 < n ? (1u << c1) : 0u;
#pragma unroll
                for (int k = 0; k < RTree<NM>::NI; ++k)
                    if (k < r) { m0 = (c0 - n == k) ? below[k] : m0; m1 = (c1 - n == k) ? below[k] : m1; }
                below[r] = m0 | m1;
                if (r == rp) cut = sb ? m1 : m0;
            }
        }
        cx.last_desc = cut;
        if (cx.want_desc_new) {
#pragma unroll
            for (int r = 0; r < RTree<NM>::NI; ++r) tmask[r] = below[r];
        }
    }
    double tc = r_coalesce_up<NM, TAB>(cx, t, n - 1, n, h, u_refresh);
    *tc_out = tc;
    double Sp = t.getS(rp);
    int b_id = t.getC(rp, sb), s_id = t.getC(rp, 1 - sb);
    bool p_was_root = (rp == n - 2);
    r_remove_rank(t, n, n - 1, rp, s_id, &b_id, &s_id);
    int ni = n - 2;
    int troot = p_was_root ? s_id : n + (ni - 1);
    int pr = -1, ps = 0;
    int nslots = r_lineages_at(t, n, ni, tc, -1, &pr, &ps);
    bool has_root = tc >= r_node_h(t, n, troot);
    bool has_stub = tc < Sp;
    int k = nslots + (has_root ? 1 : 0) + (has_stub ? 1 : 0);
    double u = u_attach;
    int idx = min((int)(u * (double)k), k - 1);
    if (sp_out) *sp_out = Sp;
    if (changed_out) *changed_out = !(has_stub && idx == k - 1);
    // where the floating lineage re-attaches: one insertion for all three outcomes (under divergence three
    // separate calls would each be executed by the whole wavefront)
    double h_ins = tc;
    int pr_ins = -1, ps_ins = 0;
    if (idx < nslots) {
        r_lineages_at(t, n, ni, tc, idx, &pr_ins, &ps_ins);
    } else if (!(has_root && idx == nslots)) {
        // coalesced back into its own branch above the cut: the tree is unchanged -> restore p on the sibling
        // lineage's slot at time Sp (or above the pruned root)
        h_ins = Sp;
        if (!p_was_root) {
            int R = 0;
#pragma unroll
            for (int kk = 0; kk < RTree<NM>::NI; ++kk) R += (kk < ni && t.S[kk] <= Sp) ? 1 : 0;
            bool found = false;
#pragma unroll
            for (int rr = 0; rr < RTree<NM>::NI; ++rr) {
                if (rr >= R && rr < ni) {
                    int id0 = t.C0[rr];
                    if (!found && (id0 < n || id0 - n < R) && id0 == s_id) { found = true; pr_ins = rr; ps_ins = 0; }
                    int id1 = t.C1[rr];
                    if (!found && (id1 < n || id1 - n < R) && id1 == s_id) { found = true; pr_ins = rr; ps_ins = 1; }
                }
            }
        }
    }
    if (cx.want_desc_new) {
        // samples below the node this update creates: the cut samples plus those below the branch it lands on (masks of
        // the tree before the cut: rank q of the pruned tree was rank q, or q + 1 from the removed parent on)
        const unsigned cut = cx.last_desc;
        unsigned dn = cut;
        if (idx < nslots) {
            const int tid_ = t.getC(pr_ins, ps_ins);
            unsigned tm = tid_ < n ? (1u << tid_) : 0u;
            const int oldr = (tid_ - n) + ((tid_ - n) >= rp ? 1 : 0);
#pragma unroll
            for (int k = 0; k < RTree<NM>::NI; ++k) tm = (tid_ >= n && oldr == k) ? tmask[k] : tm;
            dn = cut | tm;
        } else if (has_root && idx == nslots) {
            dn = (1u << n) - 1u;
        }
        cx.last_desc_new = dn;
    }
    r_insert_node(t, n, ni, h_ins, b_id, pr_ins, ps_ins, troot);
    cx.Ltree = r_tree_length(t, n);
}

// particle.cpp:699-730
template <int NM>
__device__ __forceinline__ double r_tracked_len(const RTree<NM>& t, int n, unsigned present_mask) {
    double val[RTree<NM>::NI];
    double total = 0.0;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r) {
        val[r] = 0.0;
        if (r < n - 1) {
            int c0 = t.C0[r], c1 = t.C1[r];
            double sr = t.S[r];
            double l, rr, h0 = 0.0, h1 = 0.0;
            if (c0 < n) l = (present_mask >> c0) & 1u ? 0.0 : -1.0;
            else {
                l = val[0]; h0 = t.S[0];
#pragma unroll
                for (int k = 1; k < RTree<NM>::NI; ++k) if (k < r && c0 - n == k) { l = val[k]; h0 = t.S[k]; }
            }
            if (c1 < n) rr = (present_mask >> c1) & 1u ? 0.0 : -1.0;
            else {
                rr = val[0]; h1 = t.S[0];
#pragma unroll
                for (int k = 1; k < RTree<NM>::NI; ++k) if (k < r && c1 - n == k) { rr = val[k]; h1 = t.S[k]; }
            }
            if (l >= 0.0) l += sr - h0;
            if (rr >= 0.0) rr += sr - h1;
            double v;
            if (l >= 0.0 && rr >= 0.0) { total = l + rr; v = total; }
            else if (l >= 0.0) v = l;
            else v = rr;
            val[r] = v;
        }
    }
    return total;
}

// particle.cpp:625-680
template <int NM>
__device__ __forceinline__ double r_site_lik(const RTree<NM>& t, int n, double mu, unsigned one_mask, unsigned zero_mask, bool anc) {
    double m0[RTree<NM>::NI], m1[RTree<NM>::NI];
    double res0 = 0.0, res1 = 0.0;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r) {
        m0[r] = 0.0; m1[r] = 0.0;
        if (r < n - 1) {
            int c0 = t.C0[r], c1 = t.C1[r];
            double sr = t.S[r];
            double a0, a1, b0, b1, h0 = 0.0, h1 = 0.0;
            if (c0 < n) { a0 = (one_mask >> c0) & 1u ? 0.0 : 1.0; a1 = (zero_mask >> c0) & 1u ? 0.0 : 1.0; }
            else {
                a0 = m0[0]; a1 = m1[0]; h0 = t.S[0];
#pragma unroll
                for (int k = 1; k < RTree<NM>::NI; ++k) if (k < r && c0 - n == k) { a0 = m0[k]; a1 = m1[k]; h0 = t.S[k]; }
            }
            if (c1 < n) { b0 = (one_mask >> c1) & 1u ? 0.0 : 1.0; b1 = (zero_mask >> c1) & 1u ? 0.0 : 1.0; }
            else {
                b0 = m0[0]; b1 = m1[0]; h1 = t.S[0];
#pragma unroll
                for (int k = 1; k < RTree<NM>::NI; ++k) if (k < r && c1 - n == k) { b0 = m0[k]; b1 = m1[k]; h1 = t.S[k]; }
            }
            double tl = sr - h0;
            double trr = sr - h1;
            double pl = fastexp(-tl * mu);
            double pr = fastexp(-trr * mu);
            double v0 = (a0 * pl + a1 * (1 - pl)) * (b0 * pr + b1 * (1 - pr));
            double v1 = (a1 * pl + a0 * (1 - pl)) * (b1 * pr + b0 * (1 - pr));
            m0[r] = v0; m1[r] = v1;
            if (r == n - 2) { res0 = v0; res1 = v1; }
        }
    }
    double p0 = anc ? 1.0 : 0.5, p1 = anc ? 0.0 : 0.5;
    return res0 * p0 + res1 * p1;
}

// The same likelihood in two steps.  (1) The no-mutation probabilities of the 2 (n - 1) branches: they depend on the tree
// alone, not on each other and not on the data, and each is a chain of some fifty dependent f64 instructions (fastexp: two
// divisions), so they are computed side by side -- written as one straight-line block that the wavefront takes when
// every argument of every lane is in fastexp's rational range (the scheduler then interleaves the chains; inside
// fastexp's own branch each chain would run alone), lane by lane through fastexp otherwise: the same operations on the
// same numbers either way.  (2) The pruning pass per phasing configuration (r_site_lik_from), which reuses them.
template <int NM>
struct RBranchP { double pl[RTree<NM>::NI], pr[RTree<NM>::NI]; };
template <int NM>
__device__ __forceinline__ void r_site_branch_probs(const RTree<NM>& t, int n, double mu, RBranchP<NM>& bp) {
    constexpr int NI = RTree<NM>::NI;
    double x[2 * NI];
    bool small = true;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        x[2 * r] = 0.0; x[2 * r + 1] = 0.0;
        if (r < n - 1) {
            const int c0 = t.C0[r], c1 = t.C1[r];
            const double sr = t.S[r];
            double h0 = 0.0, h1 = 0.0;
#pragma unroll
            for (int k = 0; k < NI; ++k) if (k < r) { h0 = (c0 - n == k) ? t.S[k] : h0; h1 = (c1 - n == k) ? t.S[k] : h1; }
            const double tl = sr - h0, trr = sr - h1;
            x[2 * r] = -tl * mu; x[2 * r + 1] = -trr * mu;
            small = small && (x[2 * r] * x[2 * r] < 0.516167859) && (x[2 * r + 1] * x[2 * r + 1] < 0.516167859);
        }
    }
    double e[2 * NI];
    if (__all(small)) {
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i) { const double xx = x[i] * x[i]; e[i] = 1 + 2 * x[i] / (2 - x[i] + xx / (6 + xx * 0.1)); }
    } else {
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i) e[i] = fastexp(x[i]);
    }
#pragma unroll
    for (int r = 0; r < NI; ++r) { bp.pl[r] = e[2 * r]; bp.pr[r] = e[2 * r + 1]; }
}
template <int NM>
__device__ __forceinline__ double r_site_lik_from(const RTree<NM>& t, int n, const RBranchP<NM>& bp, unsigned one_mask, unsigned zero_mask, bool anc) {
    double m0[RTree<NM>::NI], m1[RTree<NM>::NI];
    double res0 = 0.0, res1 = 0.0;
#pragma unroll
    for (int r = 0; r < RTree<NM>::NI; ++r) {
        m0[r] = 0.0; m1[r] = 0.0;
        if (r < n - 1) {
            const int c0 = t.C0[r], c1 = t.C1[r];
            double a0, a1, b0, b1;
            if (c0 < n) { a0 = (one_mask >> c0) & 1u ? 0.0 : 1.0; a1 = (zero_mask >> c0) & 1u ? 0.0 : 1.0; }
            else {
                a0 = m0[0]; a1 = m1[0];
#pragma unroll
                for (int k = 1; k < RTree<NM>::NI; ++k) if (k < r && c0 - n == k) { a0 = m0[k]; a1 = m1[k]; }
            }
            if (c1 < n) { b0 = (one_mask >> c1) & 1u ? 0.0 : 1.0; b1 = (zero_mask >> c1) & 1u ? 0.0 : 1.0; }
            else {
                b0 = m0[0]; b1 = m1[0];
#pragma unroll
                for (int k = 1; k < RTree<NM>::NI; ++k) if (k < r && c1 - n == k) { b0 = m0[k]; b1 = m1[k]; }
            }
            const double pl = bp.pl[r], pr = bp.pr[r];
            double v0 = (a0 * pl + a1 * (1 - pl)) * (b0 * pr + b1 * (1 - pr));
            double v1 = (a1 * pl + a0 * (1 - pl)) * (b1 * pr + b0 * (1 - pr));
            m0[r] = v0; m1[r] = v1;
            if (r == n - 2) { res0 = v0; res1 = v1; }
        }
    }
    double p0 = anc ? 1.0 : 0.5, p1 = anc ? 0.0 : 0.5;
    return res0 * p0 + res1 * p1;
}

}  // namespace pf
