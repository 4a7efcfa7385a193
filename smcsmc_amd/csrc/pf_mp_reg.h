// smcsmc_amd/csrc/pf_mp_reg.h -- structured models on the register-resident local tree (n <= 8).
//
// The same update as pf_mp.h (reference: /root/reference/src/particle.cpp:1266-1521 for the control flow, 251-300 for
// the event records; same arithmetic, draw order and enumeration order as the oracle's "structured models" section),
// with the storage of pf_tree_reg.h: node heights and child ids in VGPRs, and the small per-node integers of the
// structured model packed two bits each into one register --
//     pn   population of the coalescent node of rank r            (bits 2r, 2r+1)
//     bp   walk scratch: population of the lineage above node id   (bits 2id, 2id+1; ids < 2 * 8 - 1)
//     sp   population of sample i                                  (the model's, the same in every lane)
// so that the coalescence walk reads LDS only for the epoch tables and for the migration events of the tree (a list
// per lane, sorted by time, which stays in LDS: its length is not bounded by n).  With one wavefront per SIMD the LDS
// tree of pf_mp.h spends its time in dependent LDS round trips (structured kernels).
#pragma once
#include "pf_tree_reg.h"
#include "pf_mp.h"

namespace pf {

struct MRLane {
    double* Mt;        // &sMt[tid]   migration events of this lane's tree: time ...
    int8_t* Mb;        // &sMb[tid]   ... branch (node id below it, or a PF_TAG_* while an update is in flight)
    int8_t* Mq;        // &sMq[tid]   ... population the lineage moves to
    int nm;
    int mcap;          // capacity of the event list (KArgs::mcap)
    int P;
    unsigned pn, bp, sp;
    unsigned long long nep;   // epoch of the coalescent node of rank r, six bits each (kept with the tree: no search per walk)
    const double* I2;  // [E*P]   1/(2 N_e,p)                 (LDS)
    const double* MR;  // [E*P*P] migration rates p -> q      (LDS)
    const double* MT;  // [E*P]   total emigration rate       (LDS)
    const double* CI;  // [E*P]   cumulative coalescence intensity at the epoch starts  (LDS)
    const double* CM;  // [E*P]   cumulative emigration intensity at the epoch starts   (LDS)
    const double* TJ;  // [E]     start of the next epoch with a fixed-time move, +inf if none (LDS)
    const int* EJ;     // [E]     that epoch (E if none)                                       (LDS)
    const int* JM;     // [E*P]   fixed-time moves at the start of epoch e (LDS)
    const double* vbm; // [E*P*P] variational-Bayes factor of a migration event (global), or null
    int err;           // as MLane::err
};

__device__ __forceinline__ int pk2_get(unsigned v, int i) { return (int)((v >> (2 * i)) & 3u); }
__device__ __forceinline__ unsigned pk2_set(unsigned v, int i, int x) { return (v & ~(3u << (2 * i))) | ((unsigned)x << (2 * i)); }
// entry i leaves, the entries above it move down / a new entry i arrives, the entries from i on move up (i <= 6)
__device__ __forceinline__ unsigned pk2_remove(unsigned v, int i) {
    const unsigned low = v & ((1u << (2 * i)) - 1u);
    return low | ((v >> (2 * i + 2)) << (2 * i));
}
__device__ __forceinline__ unsigned pk2_insert(unsigned v, int i, int x) {
    const unsigned low = v & ((1u << (2 * i)) - 1u);
    return low | ((unsigned)x << (2 * i)) | ((v >> (2 * i)) << (2 * i + 2));
}
// the same with six bits per entry (node epochs)
__device__ __forceinline__ int pk6_get(unsigned long long v, int i) { return (int)((v >> (6 * i)) & 63ull); }
__device__ __forceinline__ unsigned long long pk6_remove(unsigned long long v, int i) {
    const unsigned long long low = v & ((1ull << (6 * i)) - 1ull);
    return low | ((v >> (6 * i + 6)) << (6 * i));
}
__device__ __forceinline__ unsigned long long pk6_insert(unsigned long long v, int i, int x) {
    const unsigned long long low = v & ((1ull << (6 * i)) - 1ull);
    return low | ((unsigned long long)x << (6 * i)) | ((v >> (6 * i)) << (6 * i + 6));
}
// epochs of all node heights of a tree: one batch of four-way searches (independent chains of LDS reads)
template <int NM>
__device__ __forceinline__ unsigned long long rmp_node_epochs(const RCtx& cx, const RTree<NM>& t) {
    constexpr int NI = RTree<NM>::NI;
    double tv[NI];
    int ev[NI];
#pragma unroll
    for (int r = 0; r < NI; ++r) tv[r] = t.S[r];
    r_search4_batch<NI>(cx.T, tv, ev);
    unsigned long long nep = 0;
#pragma unroll
    for (int r = 0; r < NI; ++r) nep |= (unsigned long long)ev[r] << (6 * r);
    return nep;
}

// samples below the branch above child sb of rank rp (get_descendants, descendants.hpp:22-33): masks bottom-up
template <int NM>
__device__ __forceinline__ unsigned r_desc_mask(const RTree<NM>& t, int n, int rp, int sb) {
    constexpr int NI = RTree<NM>::NI;
    unsigned below[NI];
    unsigned cut = 0;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        below[r] = 0;
        if (r < n - 1) {
            const int c0 = t.C0[r], c1 = t.C1[r];
            unsigned m0 = c0 < n ? (1u << c0) : 0u, m1 = c1 < n ? (1u << c1) : 0u;
#pragma unroll
            for (int k = 0; k < NI; ++k)
                if (k < r) { m0 = (c0 - n == k) ? below[k] : m0; m1 = (c1 - n == k) ? below[k] : m1; }
            below[r] = m0 | m1;
            if (r == rp) cut = sb ? m1 : m0;
        }
    }
    return cut;
}

// samplePoint without weights (particle.cpp:1020-1050 with one band): uniform on the tree length, slot in canonical order
template <int NM>
__device__ __forceinline__ void r_sample_point_plain(RCtx& cx, const RTree<NM>& t, double u_point, double* h_out, int* lin_out) {
    const int n = cx.n;
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
    const double q = sel_r / sel_d;
    const int lin = min((int)q, sel_k - 1);
    double h = sel_prev + (q - (double)lin) * sel_d;
    if (!(h < sel_sr)) h = sel_prev;
    *h_out = h;
    *lin_out = lin;
}

// mp_slots_at of pf_mp.h: lineages in population `pop` crossing tc in the canonical order of the tree without its node
// of rank rp; populations from bp as the walk left them.  S ascends, so "node id is at or below tc" is a rank test.
template <int NM>
__device__ __forceinline__ int rmp_slots_at(const RTree<NM>& t, const MRLane& ml, int n, int rp, double tc, int pop, int s_id, double Sp,
                                            int want, int* pr, int* ps, int* eff_out = nullptr) {
    constexpr int NI = RTree<NM>::NI;
    const int ni = n - 1;
    const int pid = n + rp;
    int R = 0;
#pragma unroll
    for (int k = 0; k < NI; ++k) R += (k < ni && t.S[k] <= tc) ? 1 : 0;
    const bool s_below = s_id < n || s_id - n < R;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        if (r < ni && r != rp && r >= R) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int id = s ? t.C1[r] : t.C0[r];
                int eff = id;
                bool crossing;
                if (id == pid) {
                    if (tc >= Sp) crossing = true;
                    else { eff = s_id; crossing = s_below; }
                } else {
                    crossing = id < n || id - n < R;
                }
                if (crossing && pk2_get(ml.bp, eff) == pop) {
                    if (cnt == want) { *pr = r - (r > rp ? 1 : 0); *ps = s; if (eff_out) *eff_out = eff; }
                    ++cnt;
                }
            }
        }
    }
    return cnt;
}

#define PF_MPR_FLUSH_BUFFER()                                                                                    \
    do {                                                                                                        \
        if (nb > 0) { mp_ev_insert(ml, bt0, bg0, bq0); ++j; } /* lands at or before j: j keeps its event */     \
        if (nb > 1) { mp_ev_insert(ml, bt1, bg1, bq1); ++j; }                                                   \
        if (nb > 2) { mp_ev_insert(ml, bt2, bg2, bq2); ++j; }                                                   \
        if (nb > 3) { mp_ev_insert(ml, bt3, bg3, bq3); ++j; }                                                   \
        nb = 0;                                                                                                 \
        fetch_event();                                                                                          \
    } while (0)
#define PF_MPR_BUF_PUSH(T_, TAG_, Q_)                                                                           \
    do {                                                                                                        \
        if (nb == 0) { bt0 = (T_); bg0 = (TAG_); bq0 = (Q_); }                                                  \
        else if (nb == 1) { bt1 = (T_); bg1 = (TAG_); bq1 = (Q_); }                                             \
        else if (nb == 2) { bt2 = (T_); bg2 = (TAG_); bq2 = (Q_); }                                             \
        else { bt3 = (T_); bg3 = (TAG_); bq3 = (Q_); }                                                          \
        ++nb;                                                                                                   \
    } while (0)

// mp_coalesce of pf_mp.h for the full tree: the lineage cut from the branch above node b_id at height h moves up.
// Leaves the populations of all lineages at the coalescence time in ml.bp.
template <int NM, bool LOG>
__device__ __forceinline__ void rmp_coalesce(RCtx& cx, const RTree<NM>& t, MRLane& ml, int b_id, double h, PLog& pl, MWalk& W) {
    constexpr int NI = RTree<NM>::NI;
    MP_TICK(tw0);
    const int P = ml.P;
    const int n = cx.n;
    const int ni = n - 1;
    const double Hr = t.getS(n - 2);
    double tt = h;
    int e = r_search4(cx.T, tt);
    int i = 0, j = 0;
#pragma unroll
    for (int k = 0; k < NI; ++k) i += (k < ni && t.S[k] <= tt) ? 1 : 0;
    W.tfirst = -1.0;
    W.e0 = e; W.e1 = cx.E - 1;
    if (LOG) { pl.fopen = false; pl.ropen = false; }
    // Every stretch of the walk ends at a node, at an event of the tree or at a fixed-time move, and needs the epoch of
    // its end: the nodes' epochs travel with the tree (ml.nep), events carry theirs, the moves' come from a table.
    const unsigned long long nep = ml.nep;
    // population of the lineage above every node id: the samples', the nodes' own, then the events up to the cut
    unsigned bp = ml.sp | (ml.pn << (2 * n));
    double nS = PF_INF, eT = PF_INF;
    int nC0 = 0, nC1 = 0, nP = 0, nE = 0, eB = 0, eQ = 0, eE = 0;
    auto fetch_node = [&]() __attribute__((always_inline)) {
        if (i < ni) { nS = t.getS(i); nC0 = t.getC(i, 0); nC1 = t.getC(i, 1); nP = pk2_get(ml.pn, i); nE = (int)((nep >> (6 * i)) & 63u); }
        else nS = PF_INF;
    };
    auto fetch_event = [&]() __attribute__((always_inline)) {
        if (j < ml.nm) { eT = LMt(ml, j); eB = LMb(ml, j); const int by = (unsigned char)LMq(ml, j); eQ = by & 3; eE = by >> 2; }
        else eT = PF_INF;
    };
    fetch_node();
    fetch_event();
    while (eT <= tt) {
        if (eB < PF_TAG_MIN) bp = pk2_set(bp, eB, eQ);
        ++j;
        fetch_event();
    }
    int pf = pk2_get(bp, b_id), pr = pk2_get(ml.pn, n - 2);
    // lineages of the stored tree per population (eight bits each), kept up to date while the walk moves up
    unsigned cnt = 0;
#pragma unroll
    for (int r = 0; r < NI; ++r)
        if (r >= i && r < ni) {
            const int id0 = t.C0[r], id1 = t.C1[r];
            if (id0 < n || id0 - n < i) cnt += 1u << (8 * pk2_get(bp, id0));
            if (id1 < n || id1 - n < i) cnt += 1u << (8 * pk2_get(bp, id1));
        }
    auto advance = [&](double tnew) __attribute__((always_inline)) {
        while (nS <= tnew) {
            cnt -= 1u << (8 * pk2_get(bp, nC0));
            cnt -= 1u << (8 * pk2_get(bp, nC1));
            if (i < ni - 1) cnt += 1u << (8 * nP);       // the top node's own lineage is the root lineage, not a slot
            ++i;
            fetch_node();
        }
        while (eT <= tnew) {
            if (eB < PF_TAG_MIN) {
                cnt -= 1u << (8 * pk2_get(bp, eB));
                bp = pk2_set(bp, eB, eQ);
                cnt += 1u << (8 * eQ);
            }
            ++j;
            fetch_event();
        }
    };
    auto record = [&](bool root_active, int weight, double t0, double t1, int kind, int to) __attribute__((always_inline)) {
        if (!LOG) return;
        if (!pl.on) return;
        if (pl.fopen && (pl.fw != weight || pl.fp != pf)) plog_flush_f(pl, 0, 0);
        if (!pl.fopen) { pl.fopen = true; pl.fw = weight; pl.fp = pf; pl.ft0 = t0; }
        pl.ft1 = t1;
        if (kind == 1) plog_flush_f(pl, 1, 0);
        if (kind == 2) plog_flush_f(pl, 2, to);
        if (root_active) {
            if (pl.ropen && pl.rp != pr) plog_flush_r(pl, 0, 0);
            if (!pl.ropen) { pl.ropen = true; pl.rp = pr; pl.rt0 = t0; }
            pl.rt1 = t1;
            if (kind == 3) plog_flush_r(pl, 2, to);
        }
    };
    auto ci = [&](int q, int ee, double tm) __attribute__((always_inline)) { return ml.CI[ee * P + q] + (tm - cx.T[ee]) * ml.I2[ee * P + q]; };
    auto cm = [&](int q, int ee, double tm) __attribute__((always_inline)) { return ml.CM[ee * P + q] + (tm - cx.T[ee]) * ml.MT[ee * P + q]; };
    double bt0 = 0, bt1 = 0, bt2 = 0, bt3 = 0;
    int bq0 = 0, bq1 = 0, bq2 = 0, bq3 = 0, bg0 = 0, bg1 = 0, bg2 = 0, bg3 = 0;
    int nb = 0;
    MP_TICK(tw1);
    MP_ACC(ml, 2, tw0, tw1);
    bool done = false;
    // cumulative intensities at the start of the stretch: what the end of the last stretch computed, unless a lineage
    // changed population or the root's lineage has just become active
    double f0c = 0.0, f0m = 0.0, f0r = 0.0;
    bool f_fresh = false, f_root = false;
    // One trip of the outer loop per EVENT of the walk (a migration, or the coalescence that ends it), in two phases that
    // every lane of the wavefront goes through together:
    //   1. stretches without an event -- a short loop (boundary, hazard, compare, advance) that a lane leaves when its
    //      budget runs out in the stretch at hand; the lanes wait for the one with the most stretches to pass;
    //   2. the event -- its epoch, one division, the kind, the record -- once, with the wavefront converged.
    // The same arithmetic in the same order as one loop over stretches with the event code inside; but there a wavefront
    // whose lanes are in different places pays for the event code in nearly every iteration (it was two fifths of the
    // iteration), here once per event.  The event's two random numbers are drawn at the top, converged as well.
    for (int guard = 0; guard < 4096 && !done; ++guard) {
        MP_CYC(cy_o0);
        if (nb > 1) PF_MPR_FLUSH_BUFFER();
        const double u_type = philox_uniform(cx.seed, cx.slot, cx.stream, cx.ctr);
        const double eb_new = -dlog(philox_uniform(cx.seed, cx.slot, cx.stream, cx.ctr + 1));
        MP_CYC(cy_o1);
        MP_ACC(ml, 16, cy_o0, cy_o1);
        bool root_active = false;
        double tn = PF_INF;
        int weight = 0, en = e;
        // ---- phase 1
        for (int g2 = 0; g2 < 100000; ++g2) {
            MP_ACC(ml, 13, 0, 1);
            MP_CYC(cy0);
            if (nb > 1) PF_MPR_FLUSH_BUFFER();            // joins may have queued events (rare)
            root_active = tt >= Hr;
            const double tj = ml.TJ[e];
            tn = nS < eT ? nS : eT;
            tn = tn < tj ? tn : tj;
            const int k = (int)((cnt >> (8 * pf)) & 0xffu);
            weight = k + ((root_active && pr == pf) ? 1 : 0);
            en = e;
            if (!f_fresh || f_root != root_active) {
                f0c = ci(pf, e, tt); f0m = cm(pf, e, tt); f0r = root_active ? cm(pr, e, tt) : 0.0;
                f_fresh = true; f_root = root_active;
            }
            if (!(tn < PF_INF)) break;                    // nothing above but the event
            en = !(nS > tn) ? nE : (!(eT > tn) ? eE : ml.EJ[e]);
            const double f1c = ci(pf, en, tn), f1m = cm(pf, en, tn);
            double f1r = 0.0;
            double need = (double)weight * (f1c - f0c) + (f1m - f0m);
            if (root_active) { f1r = cm(pr, en, tn); need = need + (f1r - f0r); }
            MP_CYC(cy1);
            MP_ACC(ml, 17, cy0, cy1);
            if (!(cx.ebuf > need)) break;                 // the event falls into this stretch
            cx.ebuf -= need;
            // the configuration changes at tn: node or event of the stored tree, or a fixed-time move
            record(root_active, weight, tt, tn, 0, 0);
            const bool at_join = !(tn < tj);
            tt = tn;
            e = en;
            f0c = f1c; f0m = f1m; f0r = f1r;              // ci(pf, en, tn) is ci(pf, e, tt) of the next stretch
            advance(tt);
            if (at_join) {
                int q = ml.JM[e * P + pf];
                if (q != pf) { PF_MPR_BUF_PUSH(tt, PF_TAG_PATH, mp_ev_byte(q, e)); pf = q; f_fresh = false; }
                if (tt >= Hr) {
                    int qr = ml.JM[e * P + pr];
                    if (qr != pr) { PF_MPR_BUF_PUSH(tt, PF_TAG_RPATH, mp_ev_byte(qr, e)); pr = qr; f_fresh = false; }
                }
            }
            MP_CYC(cy5);
            MP_ACC(ml, 20, cy1, cy5);
            if (ml.err) break;
        }
        if (ml.err) { ml.bp = bp; W.tc = tt; W.pf = pf; W.pr = pr; W.weight = 0; return; }
        // ---- phase 2: the event, in the stretch [tt, tn)
        MP_CYC(cy2a);
        {
            // its epoch: the number of epoch starts of the stretch the budget still reaches (see mp_coalesce), counted
            // four at a time with their reads in flight together
            const int elim = tn < PF_INF ? en : cx.E - 1;
            int ee = e;
            double gee = 0.0;
            while (ee < elim) {
                double g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = ee + 1 + q < cx.E ? ee + 1 + q : cx.E - 1;       // past the stretch: read, never used
                    g[q] = (double)weight * (ml.CI[kk * P + pf] - f0c) + (ml.CM[kk * P + pf] - f0m);
                    if (root_active) g[q] = g[q] + (ml.CM[kk * P + pr] - f0r);
                }
                int adv = 0;
                bool open = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    open = open && ee + 1 + q <= elim && cx.ebuf > g[q];
                    if (open) { gee = g[q]; ++adv; }
                }
                ee += adv;
                if (adv < 4) break;
            }
            const double rc = (double)weight * ml.I2[ee * P + pf];
            const double rmf = ml.MT[ee * P + pf];
            const double rmr = root_active ? ml.MT[ee * P + pr] : 0.0;
            const double lam = (rc + rmf) + rmr;
            if (lam == 0.0) { if (!ml.err) ml.err = 3; ml.bp = bp; W.tc = tt; W.pf = pf; W.pr = pr; W.weight = 0; return; }
            double t1 = ee == e ? tt + cx.ebuf / lam : cx.T[ee] + (cx.ebuf - gee) / lam;
            {
                double up = r_epoch_end(cx, ee);
                up = up < tn ? up : tn;
                if (t1 > up) t1 = up;
            }
            if (W.tfirst < 0.0) W.tfirst = t1;
            int kind, to = 0;
            {
                double v = u_type * lam;
                if (v < rc || (rmf == 0.0 && rmr == 0.0)) kind = 1;
                else {
                    v -= rc;
                    int from;
                    if (v < rmf || rmr == 0.0) { kind = 2; from = pf; }
                    else { kind = 3; from = pr; v -= rmf; }
                    to = -1;
                    for (int q = 0; q < P; ++q) {
                        double mr = ml.MR[(ee * P + from) * P + q];
                        if (q == from || mr == 0.0) continue;
                        to = q;
                        if (v < mr) break;
                        v -= mr;
                    }
                }
            }
            record(root_active, weight, tt, t1, kind, to);
            if (cx.vbc) cx.upd_fac *= kind == 1 ? cx.vbc[ee * P + pf] : ml.vbm[(ee * P + (kind == 2 ? pf : pr)) * P + to];
            cx.ebuf = eb_new;
            cx.ctr += 2;
            if (kind == 1) {
                W.tc = t1; W.pf = pf; W.pr = pr; W.weight = weight;
                W.e1 = (ee + 1 < cx.E && !(t1 < cx.T[ee + 1])) ? ee + 1 : ee;
                done = true;
            } else {
                PF_MPR_BUF_PUSH(t1, kind == 2 ? PF_TAG_PATH : PF_TAG_RPATH, mp_ev_byte(to, ee));
                if (kind == 2) pf = to; else pr = to;
                tt = t1;
                e = ee;
                f_fresh = false;
            }
        }
        MP_CYC(cy3);
        MP_ACC(ml, 18, cy2a, cy3);
        MP_ACC(ml, 12, 0, 1);
    }
    ml.bp = bp;
    if (!done) { if (!ml.err) ml.err = 3; W.tc = tt; W.pf = pf; W.pr = pr; W.weight = 0; return; }
    MP_TICK(tw2);
    MP_ACC(ml, 3, tw1, tw2);
    PF_MPR_FLUSH_BUFFER();
    if (LOG) { plog_flush_f(pl, 0, 0); plog_flush_r(pl, 0, 0); }
    MP_TICK(tw3);
    MP_ACC(ml, 4, tw2, tw3);
}
#undef PF_MPR_FLUSH_BUFFER
#undef PF_MPR_BUF_PUSH

// mp_genealogy_rest of pf_mp.h: the update after the recombination point (slot (rp, sb), height h) has been sampled
// TREES (-arg): *desc_new receives the samples below the node the update creates -- the cut samples `cut` plus those
// below the lineage it lands on, all of them above the root, its own only when it falls back into its branch
template <int NM, bool LOG, bool TREES = false>
__device__ __forceinline__ void rmp_genealogy_rest(RCtx& cx, RTree<NM>& t, MRLane& ml, PLog& pl, int rp, int sb, double h,
                                                   double* tc_out, double* tfirst_out, unsigned cut = 0, unsigned* desc_new = nullptr) {
    constexpr int NI = RTree<NM>::NI;
    const int n = cx.n;
    unsigned below[NI];
    if (TREES) {
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            below[r] = 0;
            if (r < n - 1) {
                const int c0 = t.C0[r], c1 = t.C1[r];
                unsigned m0 = c0 < n ? (1u << c0) : 0u, m1 = c1 < n ? (1u << c1) : 0u;
#pragma unroll
                for (int k = 0; k < NI; ++k)
                    if (k < r) { m0 = (c0 - n == k) ? below[k] : m0; m1 = (c1 - n == k) ? below[k] : m1; }
                below[r] = m0 | m1;
            }
        }
    }
    int b_id = t.getC(rp, sb), s_id = t.getC(rp, 1 - sb);
    MWalk W;
    rmp_coalesce<NM, LOG>(cx, t, ml, b_id, h, pl, W);
    MP_TICK(tg2);
    const double tc = W.tc;
    *tc_out = tc;
    *tfirst_out = W.tfirst;
    const double Sp = t.getS(rp);
    if (ml.err) return;
    const int p_pop = pk2_get(ml.pn, rp);
    const bool p_was_root = (rp == n - 2);
    int pr = -1, ps = 0;
    const int nslots = rmp_slots_at(t, ml, n, rp, tc, W.pf, s_id, Sp, -1, &pr, &ps);
    bool has_root;
    if (p_was_root) has_root = tc >= r_node_h(t, n, s_id) && (tc < Sp ? pk2_get(ml.bp, s_id) : W.pr) == W.pf;
    else has_root = tc >= t.getS(n - 2) && W.pr == W.pf;
    const bool has_stub = tc < Sp && pk2_get(ml.bp, b_id) == W.pf;
    const int k = nslots + (has_root ? 1 : 0) + (has_stub ? 1 : 0);
    if (k != W.weight || k < 1) { if (!ml.err) ml.err = 2; return; }
    const double u = r_uni(cx);
    const int idx = min((int)(u * (double)k), k - 1);
    int eff = 0;
    if (idx < nslots) rmp_slots_at(t, ml, n, rp, tc, W.pf, s_id, Sp, idx, &pr, &ps, &eff);
    if (!TREES && desc_new) *desc_new = piece_span(W.e0, W.e1);
    if (TREES) {
        unsigned dn = cut;
        if (idx < nslots) {
            unsigned tm = eff < n ? (1u << eff) : 0u;
#pragma unroll
            for (int k = 0; k < NI; ++k) tm = (eff - n == k) ? below[k] : tm;
            dn = cut | tm;
        } else if (has_root && idx == nslots) {
            dn = (1u << n) - 1u;
        }
        *desc_new = dn;
    }
    MP_TICK(tg3);
    MP_ACC(ml, 5, tg2, tg3);
    // ---- the edit (see mp_genealogy_rest: tree by one removal and one insertion, event list in one pass)
    const int pid = n + rp;
    const int b0 = b_id, s0 = s_id;
    const bool into_stub = !(idx < nslots) && !(has_root && idx == nslots);
    ml.pn = pk2_remove(ml.pn, rp);
    const int p_epoch = pk6_get(ml.nep, rp);
    ml.nep = pk6_remove(ml.nep, rp);
    r_remove_rank(t, n, n - 1, rp, s_id, &b_id, &s_id);
    const int ni = n - 2;
    const int troot = p_was_root ? s0 : n + (ni - 1);
    double h_ins = tc;
    int pr_ins = -1, ps_ins = 0, pop_ins = W.pf;
    if (idx < nslots) { pr_ins = pr; ps_ins = ps; }
    if (into_stub) {
        h_ins = Sp; pop_ins = p_pop;
        if (!p_was_root) {
            int R = 0;
#pragma unroll
            for (int kk = 0; kk < NI; ++kk) R += (kk < ni && t.S[kk] <= Sp) ? 1 : 0;
            bool found = false;
#pragma unroll
            for (int rr = 0; rr < NI; ++rr) {
                if (rr >= R && rr < ni) {
                    const int id0 = t.C0[rr];
                    if (!found && (id0 < n || id0 - n < R) && id0 == s0) { found = true; pr_ins = rr; ps_ins = 0; }
                    const int id1 = t.C1[rr];
                    if (!found && (id1 < n || id1 - n < R) && id1 == s0) { found = true; pr_ins = rr; ps_ins = 1; }
                }
            }
        }
    }
    int rn = 0;
#pragma unroll
    for (int kk = 0; kk < NI; ++kk) rn += (kk < ni && t.S[kk] <= h_ins) ? 1 : 0;
    const int nid = n + rn;
    int tg = pr_ins >= 0 ? t.getC(pr_ins, ps_ins) : troot;
    if (tg >= nid) tg += 1;
    const int troot2 = troot >= nid ? troot + 1 : troot;
    const int b2 = b0 >= nid ? b0 + 1 : b0;
    const int root_final = n + n - 2;
    {
        int o = 0;
        const int nmv = ml.nm;
        for (int q = 0; q < nmv; ++q) {
            int v = LMb(ml, q);
            const double tm = LMt(ml, q);
            const int8_t to = LMq(ml, q);
            bool keep = true;
            if (v < PF_TAG_MIN) {
                if (v == b0 && tm > h) {
                    keep = into_stub && tm > tc;
                    v = b2;
                } else {
                    if (v == pid) v = s0; else if (v > pid) v -= 1;
                    if (v >= nid) v += 1;
                    if (v == tg && tm > h_ins) { if (pr_ins >= 0) v = nid; else keep = false; }
                }
            } else if (v == PF_TAG_RPATH) {
                v = troot2;
                if (v == tg && tm > h_ins) { if (pr_ins >= 0) v = nid; else keep = false; }
            } else {
                v = b2;
            }
            if (v == root_final) keep = false;
            if (keep) {
                LMt(ml, o) = tm; LMb(ml, o) = (int8_t)v; LMq(ml, o) = to;
                ++o;
            }
        }
        ml.nm = o;
    }
    ml.pn = pk2_insert(ml.pn, rn, pop_ins);
    ml.nep = pk6_insert(ml.nep, rn, into_stub ? p_epoch : W.e1);
    r_insert_node(t, n, ni, h_ins, b0, pr_ins, ps_ins, troot);
    MP_TICK(tg4);
    MP_ACC(ml, 6, tg3, tg4);
    cx.Ltree = r_tree_length(t, n);
}

}  // namespace pf
