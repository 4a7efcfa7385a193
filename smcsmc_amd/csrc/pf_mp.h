// smcsmc_amd/csrc/pf_mp.h -- structured models (more than one population): migration, population-specific
// coalescence rates and fixed-time population moves for the per-lane local tree held in LDS.
//
// Reference: the scrm fork's Forest::sampleCoalescences is absent; the control flow follows its
// mirror in /root/reference/src/particle.cpp:1266-1521 (sampleNextGenealogyWithoutImplementing + dontImplement*),
// the event records particle.cpp:251-300.  Same arithmetic, draw order and enumeration order as the CPU oracle
// restatement under oracle/ ("structured models"), which the parity tests compare against.
//
// Representation: the rank-sorted binary tree of pf_device.h is kept as it is; every coalescent node also
// stores the population it happened in (Pn), and the migration events of the local tree live in one list
// sorted by time: event m sits on the branch above node id Mb[m] and moves that lineage to population Mq[m] & 3
// at time Mt[m]; the upper six bits of Mq[m] hold the epoch of the event (the walk needs it at every event it passes).  Likelihood, branch-length and recombination-opportunity code never see the events.
// While a genealogy update is in flight, the events picked up by the floating lineage, by the root's own
// lineage and the events of the cut branch's stub carry the temporary branch tags below.
#pragma once
#include "pf_device.h"
#include "pf_mp_host.h"

#define PF_TAG_PATH 120       // picked up by the floating lineage during this update
#define PF_TAG_RPATH 121      // picked up by the root's lineage above the root
#define PF_TAG_STUB 122       // on the cut branch above the cut
#define PF_TAG_MIN 120

namespace pf {

// profiling builds (-DPF_STAMPS): time spent by a wavefront in the phases of the structured update, summed over the row
#ifdef PF_STAMPS
#define MP_TICK(v) const unsigned long long v = wall_clock64()
// accumulated per wavefront (one per workgroup in pf_mp.hip): the first active lane adds, so the figure covers every trip
// the wavefront makes, whichever lanes take part in it
__shared__ unsigned long long g_mp_acc[PF_STAMP_W];
__device__ __forceinline__ void mp_wave_acc(int k, unsigned long long v) {
    if (threadIdx.x < 64 && (int)threadIdx.x == __ffsll((unsigned long long)__ballot(1)) - 1) g_mp_acc[k] += v;   // first wavefront only
}
#define MP_ACC(ml, k, a, b) mp_wave_acc(k, (b) - (a))
#define MP_CYC(v) const unsigned long long v = clock64()
#else
#define MP_CYC(v) do {} while (0)
#define MP_TICK(v) do {} while (0)
#define MP_ACC(ml, k, a, b) do {} while (0)
#endif

struct MLane {
    int8_t* Pn;        // &sPn[tid]   population of coalescent node r at Pn[r*PF_BS]
    double* Mt;        // &sMt[tid]
    int8_t* Mb;        // &sMb[tid]
    int8_t* Mq;        // &sMq[tid]
    int8_t* Bp;        // &sBp[tid]   scratch of the walk: current population of the lineage above every node id
    int nm;
    int mcap;          // capacity of the event list (KArgs::mcap)
    int P;
    const double* I2;  // [E*P]   1/(2 N_e,p)                 (LDS)
    const double* MR;  // [E*P*P] migration rates p -> q      (LDS)
    const double* MT;  // [E*P]   total emigration rate       (LDS)
    const double* CI;  // [E*P]   cumulative coalescence intensity of population p at the epoch starts  (LDS)
    const double* CM;  // [E*P]   cumulative emigration intensity of population p at the epoch starts   (LDS)
    const double* TJ;  // [E]     start of the next epoch after e at which a fixed-time move (-ej) happens, +inf if none (LDS)
    const int* JM;     // [E*P]   fixed-time moves at the start of epoch e (LDS)
    const int* SP;     // [n]     sample populations          (LDS)
    const double* vbm; // [E*P*P] variational-Bayes factor of a migration event (global), or null
    int err;           // 1 list overflow, 2 partner count inconsistent, 3 no final coalescence possible
};

#define LPn(ml, r) ((ml).Pn[(r) * PF_BS])
#define LMt(ml, m) ((ml).Mt[(m) * PF_BS])
#define LMb(ml, m) ((ml).Mb[(m) * PF_BS])
#define LMq(ml, m) ((ml).Mq[(m) * PF_BS])
#define LBp(ml, id) ((ml).Bp[(id) * PF_BS])

// coal/migr opportunity pieces of one genealogy update, written to the slot's piece ring (three words each):
//   tag = pop | kind << 8 | to << 16 | weight << 24   (kind bit0: coalescence at the end, bit1: migration to `to`,
//                                                       bit2: piece of the root's own lineage, not of the floating one)
//   t0, t1 = the stretch of the walk during which the lineage sat in `pop` with `weight` coalescence partners.
// A piece may span several epochs; the count kernel clips it to the epoch it is counting (coalescence opportunity =
// weight * overlap, migration opportunity = overlap, the event belongs to the epoch that holds t1) and applies the
// record flags and the epoch limit of the row there.
struct PLog {
    double* base;            // ring of this slot
    unsigned cap;
    unsigned idx;            // pieces ever written by this slot
    unsigned pos;            // idx % cap, kept incrementally
    bool on;
    int fw, fp, rp;          // open piece of the floating lineage (weight, population) / of the root lineage (population)
    double ft0, ft1, rt0, rt1;
    bool fopen, ropen;
};

__device__ __forceinline__ void plog_write(PLog& pl, int pop, int kind, int to, int weight, double t0, double t1) {
    double* q = pl.base + (size_t)pl.pos * 3;
    long long tag = (long long)(pop & 0xff) | ((long long)(kind & 0xff) << 8) | ((long long)(to & 0xff) << 16) |
                    ((long long)(weight & 0xff) << 24);
    q[0] = __longlong_as_double(tag);
    q[1] = t0;
    q[2] = t1;
    ++pl.idx;
    if (++pl.pos == pl.cap) pl.pos = 0;
}
__device__ __forceinline__ void plog_flush_f(PLog& pl, int kind, int to) {
    if (pl.fopen) plog_write(pl, pl.fp, kind, to, pl.fw, pl.ft0, pl.ft1);
    pl.fopen = false;
}
__device__ __forceinline__ void plog_flush_r(PLog& pl, int kind, int to) {
    if (pl.ropen) plog_write(pl, pl.rp, kind | 4, to, 0, pl.rt0, pl.rt1);
    pl.ropen = false;
}

__device__ __forceinline__ int mp_pop_base(const Lane& ln, const MLane& ml, int id) {
    return id < ln.n ? ml.SP[id] : (int)LPn(ml, id - ln.n);
}
__device__ __forceinline__ int mp_pop_at(const Lane& ln, const MLane& ml, int id, double time) {
    int pop = mp_pop_base(ln, ml, id);
    for (int m = 0; m < ml.nm; ++m)
        if (LMb(ml, m) == id && LMt(ml, m) <= time) pop = LMq(ml, m) & 3;
    return pop;
}
__device__ __forceinline__ int mp_lineages_in_pop(const Lane& ln, const MLane& ml, int ni, double time, int pop, int want,
                                                  int* pr, int* ps) {
    int R = 0;
    while (R < ni && LS(ln, R) <= time) ++R;
    int cnt = 0;
    for (int r = R; r < ni; ++r)
        for (int s = 0; s < 2; ++s) {
            int id = LC(ln, r, s);
            if ((id < ln.n || id - ln.n < R) && mp_pop_at(ln, ml, id, time) == pop) {
                if (cnt == want) { *pr = r; *ps = s; }
                ++cnt;
            }
        }
    return cnt;
}
// `newpop` arrives packed: population | epoch of the event << 2 (mp_ev_byte)
__device__ __forceinline__ int mp_ev_byte(int pop, int epoch) { return pop | (epoch << 2); }
template <class ML>
__device__ __forceinline__ void mp_ev_insert(ML& ml, double time, int branch, int newpop) {
    if (ml.nm >= ml.mcap) { ml.err = 1; return; }
    int m = ml.nm;
    while (m > 0 && LMt(ml, m - 1) > time) {
        LMt(ml, m) = LMt(ml, m - 1); LMb(ml, m) = LMb(ml, m - 1); LMq(ml, m) = LMq(ml, m - 1);
        --m;
    }
    LMt(ml, m) = time; LMb(ml, m) = (int8_t)branch; LMq(ml, m) = (int8_t)newpop;
    ++ml.nm;
}
// drop the events on branch `id` later than tmin
__device__ __forceinline__ void mp_ev_drop_above(MLane& ml, int id, double tmin) {
    int o = 0;
    for (int m = 0; m < ml.nm; ++m) {
        int b = LMb(ml, m);
        double t = LMt(ml, m);
        if (b == id && t > tmin) continue;
        if (o != m) { LMt(ml, o) = t; LMb(ml, o) = (int8_t)b; LMq(ml, o) = LMq(ml, m); }
        ++o;
    }
    ml.nm = o;
}
__device__ __forceinline__ void mp_remove_rank(Lane& ln, MLane& ml, int ni, int rp, int sib, int* a, int* b) {
    const int pid = ln.n + rp;
    for (int m = 0; m < ml.nm; ++m) {
        int v = LMb(ml, m);
        if (v == pid) v = sib;
        if (v > pid && v < PF_TAG_MIN) v -= 1;
        LMb(ml, m) = (int8_t)v;
    }
    for (int r = rp; r + 1 < ni; ++r) LPn(ml, r) = LPn(ml, r + 1);
    remove_rank(ln, ni, rp, sib, a, b);
}
__device__ __forceinline__ void mp_insert_node(Lane& ln, MLane& ml, int ni, double h, int* fl, int pr, int ps, int root_id,
                                               int node_pop) {
    int rn = 0;
    while (rn < ni && LS(ln, rn) <= h) ++rn;
    const int nid = ln.n + rn;
    int target = pr >= 0 ? (int)LC(ln, pr, ps) : root_id;
    for (int m = 0; m < ml.nm; ++m) {
        int v = LMb(ml, m);
        if (v >= nid && v < PF_TAG_MIN) LMb(ml, m) = (int8_t)(v + 1);
    }
    if (target >= nid) target += 1;
    if (pr >= 0) {
        for (int m = 0; m < ml.nm; ++m)
            if (LMb(ml, m) == target && LMt(ml, m) > h) LMb(ml, m) = (int8_t)nid;
    } else {
        mp_ev_drop_above(ml, target, h);
    }
    for (int r = ni; r > rn; --r) LPn(ml, r) = LPn(ml, r - 1);
    LPn(ml, rn) = (int8_t)node_pop;
    insert_node(ln, ni, h, *fl, pr, ps, root_id);
    if (*fl >= nid) *fl += 1;
}

// Lineages in population `pop` crossing time tc, enumerated in the canonical order of the tree WITHOUT its node of
// rank rp (rp < 0: nothing removed): rows ascending, child 0 then 1, the removed node's parent slot standing for
// the sibling lineage (below the removed node) or the removed node's own branch (above it).  Populations are read
// from Bp as the walk left them at tc.  Returns the count; the want-th slot is returned in pruned-tree coordinates.
__device__ __forceinline__ int mp_slots_at(const Lane& ln, const MLane& ml, int ni, int rp, double tc, int pop, int s_id,
                                           double Sp, int want, int* pr, int* ps) {
    const int n = ln.n;
    const int pid = rp >= 0 ? n + rp : -1;
    int cnt = 0;
    for (int r = 0; r < ni; ++r) {
        if (r == rp || !(LS(ln, r) > tc)) continue;
        for (int s = 0; s < 2; ++s) {
            int id = LC(ln, r, s);
            int eff = id;
            bool crossing;
            if (id == pid) {
                if (tc >= Sp) crossing = true;
                else { eff = s_id; crossing = node_h(ln, s_id) <= tc; }
            } else {
                crossing = node_h(ln, id) <= tc;
            }
            if (crossing && LBp(ml, eff) == pop) {
                if (cnt == want) { *pr = r - ((rp >= 0 && r > rp) ? 1 : 0); *ps = s; }
                ++cnt;
            }
        }
    }
    return cnt;
}

struct MWalk { double tc; int pf, pr, weight; double tfirst; int e0, e1; };   // tfirst: first sampled event of the walk (migration or coalescence); e0, e1: epochs of its start and of its coalescence
// what a record says about where its pieces lie (bits 48-63 of the meta word when no tree dump is recorded): epochs
// outside [e0, e1] need not look at them
__device__ __forceinline__ unsigned piece_span(int e0, int e1) { return (unsigned)e0 | ((unsigned)e1 << 6) | 0x1000u; }

// The floating lineage starts at height h in population pf0 and moves up through the stored tree (ni internal
// nodes, root_id its top node or the single leaf); above the root the root's own lineage is the second active
// lineage.  Migration events picked up on the way enter the list under PF_TAG_PATH / PF_TAG_RPATH.
template <bool LOG>
__device__ __forceinline__ void mp_coalesce(Lane& ln, MLane& ml, int ni, int root_id, double h, int pf0, PLog& pl,
                                            int limit, MWalk& W) {
    MP_TICK(tw0);
    const int P = ml.P;
    const int n = ln.n;
    const double Hr = node_h(ln, root_id);
    double tt = h;
    int e = epoch_of(ln, tt);
    int i = 0, j = 0;
    int pf = pf0, pr = mp_pop_base(ln, ml, root_id);
    W.tfirst = -1.0;
    W.e0 = e; W.e1 = ln.E - 1;
    if (LOG) { pl.fopen = false; pl.ropen = false; }
    // Lineages of the stored tree per population, kept up to date while the walk moves up (the restatement
    // recounts them in every interval; the numbers are the same).  Bp[id] = current population of the lineage
    // above node id; a lineage is counted from its lower node until its parent node.
    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
    auto bump = [&](int q, int d) __attribute__((always_inline)) {
        cnt0 += q == 0 ? d : 0; cnt1 += q == 1 ? d : 0; cnt2 += q == 2 ? d : 0; cnt3 += q == 3 ? d : 0;
    };
    auto count_of = [&](int q) __attribute__((always_inline)) { return q == 0 ? cnt0 : q == 1 ? cnt1 : q == 2 ? cnt2 : cnt3; };
    for (int id = 0; id < n; ++id) LBp(ml, id) = (int8_t)ml.SP[id];
    for (int r = 0; r < ni; ++r) LBp(ml, n + r) = LPn(ml, r);
    while (i < ni && LS(ln, i) <= tt) ++i;
    while (j < ml.nm && LMt(ml, j) <= tt) {
        int b = LMb(ml, j);
        if (b < PF_TAG_MIN) LBp(ml, b) = (int8_t)(LMq(ml, j) & 3);
        ++j;
    }
    for (int r = i; r < ni; ++r)
        for (int s = 0; s < 2; ++s) {
            int id = LC(ln, r, s);
            if (id < n || id - n < i) bump(LBp(ml, id), 1);
        }
    // The next node and the next event are held in registers, fetched with independent LDS reads when i / j move,
    // so crossing a boundary costs one LDS round trip instead of a chain of dependent ones.
    double nS = PF_INF, eT = PF_INF;
    int nC0 = 0, nC1 = 0, nP = 0, eB = 0, eQ = 0;
    auto fetch_node = [&]() __attribute__((always_inline)) {
        if (i < ni) { nS = LS(ln, i); nC0 = LC(ln, i, 0); nC1 = LC(ln, i, 1); nP = LPn(ml, i); }
        else nS = PF_INF;
    };
    auto fetch_event = [&]() __attribute__((always_inline)) {
        if (j < ml.nm) { eT = LMt(ml, j); eB = LMb(ml, j); eQ = LMq(ml, j) & 3; }
        else eT = PF_INF;
    };
    fetch_node();
    fetch_event();
    // move the bookkeeping over every node / event boundary at or below the new time
    auto advance = [&](double tnew) __attribute__((always_inline)) {
        while (nS <= tnew) {
            int p0 = LBp(ml, nC0), p1 = LBp(ml, nC1);
            bump(p0, -1);
            bump(p1, -1);
            if (i < ni - 1) bump(nP, 1);                 // the top node's own lineage is the root lineage, not a slot
            ++i;
            fetch_node();
        }
        while (eT <= tnew) {
            if (eB < PF_TAG_MIN) {
                bump(LBp(ml, eB), -1);
                LBp(ml, eB) = (int8_t)eQ;
                bump(eQ, 1);
            }
            ++j;
            fetch_event();
        }
    };
    // record_all_event (particle.cpp:251-300) for the stretch [t0, t1) of the walk: kind 0 no event, 1 coalescence,
    // 2 the floating lineage migrates to `to`, 3 the root lineage migrates to `to`.  Stretches of equal weight and
    // population merge into one piece whatever epochs they span.
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
    // cumulative intensities of population q at time t inside epoch ee (piecewise linear, tabulated at the epoch starts)
    auto ci = [&](int q, int ee, double t) __attribute__((always_inline)) { return ml.CI[ee * P + q] + (t - ln.T[ee]) * ml.I2[ee * P + q]; };
    auto cm = [&](int q, int ee, double t) __attribute__((always_inline)) { return ml.CM[ee * P + q] + (t - ln.T[ee]) * ml.MT[ee * P + q]; };
    // Events picked up by the two active lineages wait in registers until the walk is over (they are not lineages of
    // the stored tree, so nothing in the walk reads them); inserting into the LDS list inside the loop would make
    // the whole wavefront pay for every lane's migration.
    // (four scalar slots, not an array: a dynamically indexed array would be placed in scratch memory)
    double bt0 = 0, bt1 = 0, bt2 = 0, bt3 = 0;
    int bq0 = 0, bq1 = 0, bq2 = 0, bq3 = 0, bg0 = 0, bg1 = 0, bg2 = 0, bg3 = 0;
    int nb = 0;
#define PF_MP_FLUSH_BUFFER()                                                                                     \
    do {                                                                                                        \
        if (nb > 0) { mp_ev_insert(ml, bt0, bg0, bq0); ++j; } /* lands at or before j: j keeps its event */     \
        if (nb > 1) { mp_ev_insert(ml, bt1, bg1, bq1); ++j; }                                                   \
        if (nb > 2) { mp_ev_insert(ml, bt2, bg2, bq2); ++j; }                                                   \
        if (nb > 3) { mp_ev_insert(ml, bt3, bg3, bq3); ++j; }                                                   \
        nb = 0;                                                                                                 \
        fetch_event();                                                                                          \
    } while (0)
#define PF_MP_BUF_PUSH(T_, TAG_, Q_)                                                                            \
    do {                                                                                                        \
        if (nb == 0) { bt0 = (T_); bg0 = (TAG_); bq0 = (Q_); }                                                  \
        else if (nb == 1) { bt1 = (T_); bg1 = (TAG_); bq1 = (Q_); }                                             \
        else if (nb == 2) { bt2 = (T_); bg2 = (TAG_); bq2 = (Q_); }                                             \
        else { bt3 = (T_); bg3 = (TAG_); bq3 = (Q_); }                                                          \
        ++nb;                                                                                                   \
    } while (0)
    MP_TICK(tw1);
    MP_ACC(ml, 2, tw0, tw1);
    bool done = false;
    (void)limit;                                          // the epoch limit of the recording is applied at count time
    // A stretch lies between two changes of the configuration -- the next node of the stored tree, the next migration
    // event on it, the next epoch with a fixed-time move -- NOT between epoch boundaries: with the lineages' populations
    // and the number of partners fixed, the hazard of the two active lineages over [tt, tn) is a difference of the
    // tabulated cumulative intensities, whatever epochs lie in between (the one-population walk does the same with its
    // single table).
    // One trip of the outer loop per EVENT of the walk, in two phases the lanes of a wavefront go through together (see
    // rmp_coalesce in pf_mp_reg.h, the same loop on the register tree): the stretches without an event, then the event.
    for (int guard = 0; guard < 4096 && !done; ++guard) {
        if (nb > 1) PF_MP_FLUSH_BUFFER();
        const double u_type = philox_uniform(ln.seed, ln.slot, ln.stream, ln.ctr);
        const double eb_new = -dlog(philox_uniform(ln.seed, ln.slot, ln.stream, ln.ctr + 1));
        bool root_active = false;
        double tn = PF_INF;
        int weight = 0, en = e;
        // ---- phase 1: stretches the budget outlasts
        for (int g2 = 0; g2 < 100000; ++g2) {
            MP_ACC(ml, 13, 0, 1);
            if (nb > 1) PF_MP_FLUSH_BUFFER();             // joins may have queued events (rare)
            root_active = tt >= Hr;
            const double tj = ml.TJ[e];
            tn = nS < eT ? nS : eT;
            tn = tn < tj ? tn : tj;
            const int k = count_of(pf);
            weight = k + ((root_active && pr == pf) ? 1 : 0);
            en = e;
            if (!(tn < PF_INF)) break;                    // nothing above but the event
            while (en + 1 < ln.E && ln.T[en + 1] <= tn) ++en;
            double need = (double)weight * (ci(pf, en, tn) - ci(pf, e, tt)) + (cm(pf, en, tn) - cm(pf, e, tt));
            if (root_active) need = need + (cm(pr, en, tn) - cm(pr, e, tt));
            if (!(ln.ebuf > need)) break;                 // the event falls into this stretch
            ln.ebuf -= need;
            // the configuration changes at tn: node or event of the stored tree, or a fixed-time move
            record(root_active, weight, tt, tn, 0, 0);
            const bool at_join = !(tn < tj);
            tt = tn;
            e = en;
            advance(tt);
            if (at_join) {
                int q = ml.JM[e * P + pf];
                if (q != pf) { PF_MP_BUF_PUSH(tt, PF_TAG_PATH, mp_ev_byte(q, e)); pf = q; }
                if (tt >= Hr) {
                    int qr = ml.JM[e * P + pr];
                    if (qr != pr) { PF_MP_BUF_PUSH(tt, PF_TAG_RPATH, mp_ev_byte(qr, e)); pr = qr; }
                }
            }
            if (ml.err) break;
        }
        if (ml.err) { W.tc = tt; W.pf = pf; W.pr = pr; W.weight = 0; return; }
        // ---- phase 2: the event, in the stretch [tt, tn): its epoch is the number of epoch starts of the stretch the
        // budget still reaches (the hazard up to an epoch start is the same difference of cumulative intensities as
        // `need`, so it ascends with the epoch), then one division inside that epoch
        {
            const double f0c = ci(pf, e, tt), f0m = cm(pf, e, tt), f0r = root_active ? cm(pr, e, tt) : 0.0;
            const int elim = tn < PF_INF ? en : ln.E - 1;
            int ee = e;
            double gee = 0.0;
            while (ee < elim) {
                double g = (double)weight * (ml.CI[(ee + 1) * P + pf] - f0c) + (ml.CM[(ee + 1) * P + pf] - f0m);
                if (root_active) g = g + (ml.CM[(ee + 1) * P + pr] - f0r);
                if (!(ln.ebuf > g)) break;
                gee = g;
                ++ee;
            }
            const double rc = (double)weight * ml.I2[ee * P + pf];
            const double rmf = ml.MT[ee * P + pf];
            const double rmr = root_active ? ml.MT[ee * P + pr] : 0.0;
            const double lam = (rc + rmf) + rmr;
            if (lam == 0.0) { if (!ml.err) ml.err = 3; W.tc = tt; W.pf = pf; W.pr = pr; W.weight = 0; return; }
            double t1 = ee == e ? tt + ln.ebuf / lam : ln.T[ee] + (ln.ebuf - gee) / lam;
            {
                double up = epoch_end(ln, ee);
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
            if (ln.vbc) ln.upd_fac *= kind == 1 ? ln.vbc[ee * P + pf] : ml.vbm[(ee * P + (kind == 2 ? pf : pr)) * P + to];
            ln.ebuf = eb_new;
            ln.ctr += 2;
            if (kind == 1) {
                W.tc = t1; W.pf = pf; W.pr = pr; W.weight = weight;
                W.e1 = (ee + 1 < ln.E && !(t1 < ln.T[ee + 1])) ? ee + 1 : ee;       // an event placed at the end of its epoch counts in the next
                done = true;
            } else {
                PF_MP_BUF_PUSH(t1, kind == 2 ? PF_TAG_PATH : PF_TAG_RPATH, mp_ev_byte(to, ee));
                if (kind == 2) pf = to; else pr = to;
                tt = t1;
                e = ee;
            }
        }
        MP_ACC(ml, 12, 0, 1);
    }
    if (!done) { if (!ml.err) ml.err = 3; W.tc = tt; W.pf = pf; W.pr = pr; W.weight = 0; return; }
    MP_TICK(tw2);
    MP_ACC(ml, 3, tw1, tw2);
    PF_MP_FLUSH_BUFFER();
    if (LOG) { plog_flush_f(pl, 0, 0); plog_flush_r(pl, 0, 0); }
    MP_TICK(tw3);
    MP_ACC(ml, 4, tw2, tw3);
}
#undef PF_MP_FLUSH_BUFFER
#undef PF_MP_BUF_PUSH

__device__ __forceinline__ void mp_retag(MLane& ml, int from, int to) {
    for (int m = 0; m < ml.nm; ++m)
        if (LMb(ml, m) == from) LMb(ml, m) = (int8_t)to;
}

// Forest::buildInitialTree(true) for a structured model.  `emit(i, pstart, npieces, tc, below)` logs the record of leaf i;
// `below` = the samples under the node leaf i creates (computed when `tmp`, n-1 doubles of per-lane LDS scratch, is given;
// otherwise the epoch span of the walk's pieces, piece_span).
template <bool LOG, class Emit>
__device__ __forceinline__ void mp_build_initial_tree(Lane& ln, MLane& ml, PLog& pl, Emit emit, double* tmp = nullptr) {
    const int n = ln.n;
    ml.nm = 0;
    int root = 0;
    for (int i = 1; i < n; ++i) {
        int ni = i - 1;
        MWalk W;
        unsigned p0 = LOG ? pl.idx : 0u;
        mp_coalesce<LOG>(ln, ml, ni, root, 0.0, ml.SP[i], pl, ln.E - 1, W);
        if (ml.err) return;
        const unsigned np_ = LOG ? pl.idx - p0 : 0u;
        double tc = W.tc;
        mp_retag(ml, PF_TAG_RPATH, root);
        // candidates: the lineages of the partial tree in the coalescence population (their populations at tc are
        // what the walk left in Bp), then the root lineage
        int pr = -1, ps = 0;
        int nslots = mp_slots_at(ln, ml, ni, -1, tc, W.pf, 0, 0, -1, &pr, &ps);
        bool has_root = tc >= node_h(ln, root) && W.pr == W.pf;
        int k = nslots + (has_root ? 1 : 0);
        if (k != W.weight || k < 1) { if (!ml.err) ml.err = 2; return; }
        double u = uni(ln);
        int idx = min((int)(u * (double)k), k - 1);
        int fl = i;
        if (idx < nslots) {
            mp_slots_at(ln, ml, ni, -1, tc, W.pf, 0, 0, idx, &pr, &ps);
            emit(i, p0, np_, tc, tmp ? (1u << i) | lane_desc_mask(ln, LC(ln, pr, ps), tmp) : piece_span(W.e0, W.e1));
            mp_insert_node(ln, ml, ni, tc, &fl, pr, ps, root, W.pf);
        } else {
            emit(i, p0, np_, tc, tmp ? (2u << i) - 1u : piece_span(W.e0, W.e1));
            mp_insert_node(ln, ml, ni, tc, &fl, -1, 0, root, W.pf);
        }
        mp_retag(ml, PF_TAG_PATH, fl);
        root = n + ni;
    }
    ln.Ltree = tree_length(ln, n);
}

// the part of a genealogy update after the recombination point (slot (rp,sb), height h) has been sampled
template <bool LOG>
__device__ __forceinline__ void mp_genealogy_rest(Lane& ln, MLane& ml, PLog& pl, int limit, int rp, int sb, double h,
                                                  double* tc_out, double* sp_out, bool* changed_out, double* tfirst_out = nullptr,
                                                  unsigned* span_out = nullptr) {
    const int n = ln.n;
    MP_TICK(tg0);
    int b_id = LC(ln, rp, sb), s_id = LC(ln, rp, 1 - sb);
    const int pf0 = mp_pop_at(ln, ml, b_id, h);
    MP_TICK(tg1);
    MP_ACC(ml, 1, tg0, tg1);
    MWalk W;
    mp_coalesce<LOG>(ln, ml, n - 1, n + n - 2, h, pf0, pl, limit, W);
    MP_TICK(tg2);
    const double tc = W.tc;
    *tc_out = tc;
    if (tfirst_out) *tfirst_out = W.tfirst;
    if (span_out) *span_out = piece_span(W.e0, W.e1);
    const double Sp = LS(ln, rp);
    *sp_out = Sp;
    *changed_out = true;
    if (ml.err) return;
    const int p_pop = LPn(ml, rp);
    const bool p_was_root = (rp == n - 2);
    // candidates in the order of the pruned tree (slots, root lineage, stub), populations from the walk's Bp
    int pr = -1, ps = 0;
    const int nslots = mp_slots_at(ln, ml, n - 1, rp, tc, W.pf, s_id, Sp, -1, &pr, &ps);
    bool has_root;
    if (p_was_root) has_root = tc >= node_h(ln, s_id) && (tc < Sp ? (int)LBp(ml, s_id) : W.pr) == W.pf;
    else has_root = tc >= LS(ln, n - 2) && W.pr == W.pf;
    const bool has_stub = tc < Sp && LBp(ml, b_id) == W.pf;
    const int k = nslots + (has_root ? 1 : 0) + (has_stub ? 1 : 0);
    if (k != W.weight || k < 1) { if (!ml.err) ml.err = 2; return; }
    const double u = uni(ln);
    const int idx = min((int)(u * (double)k), k - 1);
    if (idx < nslots) mp_slots_at(ln, ml, n - 1, rp, tc, W.pf, s_id, Sp, idx, &pr, &ps);
    *changed_out = !(has_stub && idx == k - 1);
    // ---- the edit.  Tree: remove p, insert the re-attachment node (one insertion for all three outcomes: slot,
    // root lineage, back into the stub).  Event list: everything that the separate steps (tag the stub, relabel for
    // the removal, hand the root path to the pruned root, relabel for the insertion and split the target branch,
    // settle the stub, hand the floating path to the cut branch, clear the branch above the root) would do to an
    // event depends on that event alone, so it is ONE pass with in-place compaction.
    MP_TICK(tg3);
    MP_ACC(ml, 5, tg2, tg3);
    const int pid = n + rp;
    const int b0 = b_id, s0 = s_id;                      // children of p: ids below pid, unchanged by the removal
    const bool into_stub = !(idx < nslots) && !(has_root && idx == nslots);
    for (int r = rp; r + 1 < n - 1; ++r) LPn(ml, r) = LPn(ml, r + 1);
    remove_rank(ln, n - 1, rp, s_id, &b_id, &s_id);
    const int ni = n - 2;
    const int troot = p_was_root ? s0 : n + (ni - 1);
    double h_ins = tc;
    int pr_ins = -1, ps_ins = 0, pop_ins = W.pf;
    if (idx < nslots) { pr_ins = pr; ps_ins = ps; }
    if (into_stub) {
        // the tree keeps its shape: p returns at Sp on the sibling lineage's slot (or above the pruned root)
        h_ins = Sp; pop_ins = p_pop;
        if (!p_was_root) {
            int R = 0;
            while (R < ni && LS(ln, R) <= Sp) ++R;
            bool found = false;
            for (int rr = R; rr < ni && !found; ++rr)
                for (int s = 0; s < 2 && !found; ++s) {
                    int id = LC(ln, rr, s);
                    if ((id < n || id - n < R) && id == s0) { found = true; pr_ins = rr; ps_ins = s; }
                }
        }
    }
    int rn = 0;
    while (rn < ni && LS(ln, rn) <= h_ins) ++rn;
    const int nid = n + rn;
    int tg = pr_ins >= 0 ? (int)LC(ln, pr_ins, ps_ins) : troot;       // target branch (pruned-tree id) ...
    if (tg >= nid) tg += 1;                                           // ... as labelled after the insertion
    const int troot2 = troot >= nid ? troot + 1 : troot;
    const int b2 = b0 >= nid ? b0 + 1 : b0;
    const int root_final = n + n - 2;
    {
        int o = 0;
        const int nmv = ml.nm;
        for (int q = 0; q < nmv; ++q) {
            int v = LMb(ml, q);
            const double t = LMt(ml, q);
            const int8_t to = LMq(ml, q);
            bool keep = true;
            if (v < PF_TAG_MIN) {
                if (v == b0 && t > h) {
                    // the stub: events after tc return to the cut branch when the lineage fell back into its stub
                    keep = into_stub && t > tc;
                    v = b2;
                } else {
                    if (v == pid) v = s0; else if (v > pid) v -= 1;          // removal of p
                    if (v >= nid) v += 1;                                     // insertion of the new node
                    if (v == tg && t > h_ins) { if (pr_ins >= 0) v = nid; else keep = false; }
                }
            } else if (v == PF_TAG_RPATH) {
                v = troot2;
                if (v == tg && t > h_ins) { if (pr_ins >= 0) v = nid; else keep = false; }
            } else {                                                           // PF_TAG_PATH
                v = b2;
            }
            if (v == root_final) keep = false;            // nothing is kept above the root of the local tree
            if (keep) {
                LMt(ml, o) = t; LMb(ml, o) = (int8_t)v; LMq(ml, o) = to;
                ++o;
            }
        }
        ml.nm = o;
    }
    for (int r = ni; r > rn; --r) LPn(ml, r) = LPn(ml, r - 1);
    LPn(ml, rn) = (int8_t)pop_ins;
    insert_node(ln, ni, h_ins, b0, pr_ins, ps_ins, troot);
    MP_TICK(tg4);
    MP_ACC(ml, 6, tg3, tg4);
    ln.Ltree = tree_length(ln, n);
    MP_TICK(tg5);
    MP_ACC(ml, 7, tg4, tg5);
}

}  // namespace pf
