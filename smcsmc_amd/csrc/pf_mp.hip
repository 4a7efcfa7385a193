// smcsmc_amd/csrc/pf_mp.hip -- kernels for structured models (more than one population).
//
// Same structure as k_init / k_extend / k_calibrate of pf_hip.hip with the migration-aware genealogy update of
// pf_mp.h.  Compiled as its own translation unit with 64-lane workgroups: the per-lane LDS columns (local
// tree + up to PF_MMAX migration events) are large, and with at most a few hundred wavefronts per launch one
// wavefront per workgroup spreads them over more compute units.
#define PF_BS 64
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pf_device.h"
#include "pf_types.h"
#include "pf_lane.h"
#include "pf_mp.h"
#include "pf_mp_reg.h"
#include "pf_mp_host.h"
#include "pf_pipe.h"

// ------------------------------------------------------------------ structured models (P > 1): LDS-tree kernels
// Same structure as k_init / k_extend / k_calibrate with the migration-aware genealogy update of pf_mp.h.
struct SmemMP { double* I2; double* MR; double* MT; double* CI; double* CM; double* TJ; double* Mt; int* JM; int* SP; int8_t* Pn; int8_t* Mb; int8_t* Mq; int8_t* Bp; };
__host__ __device__ static size_t smem_mp_extra(int n, int E, int P, int mcap) {
    size_t dbl = (size_t)E * P * 4 + (size_t)E + (size_t)E * P * P + (size_t)mcap * PF_BS;
    size_t ints = (size_t)E * P + (size_t)((n + 1) & ~1) + (size_t)((E * P) & 1);
    size_t bytes = (size_t)(n - 1) * PF_BS + (size_t)2 * mcap * PF_BS + (size_t)2 * n * PF_BS;
    return dbl * 8 + ints * 4 + bytes;
}
static size_t smem_bytes_mp(int n, int E, int P, int mcap) { return smem_bytes(n, E) + smem_mp_extra(n, E, P, mcap); }
__device__ __forceinline__ SmemMP carve_mp(double* base, int n, int E, int P, int mcap) {
    SmemMP m;
    m.I2 = (double*)((char*)base + smem_bytes(n, E));
    m.MR = m.I2 + (size_t)E * P;
    m.MT = m.MR + (size_t)E * P * P;
    m.CI = m.MT + (size_t)E * P;
    m.CM = m.CI + (size_t)E * P;
    m.TJ = m.CM + (size_t)E * P;
    m.Mt = m.TJ + (size_t)E;
    m.JM = (int*)(m.Mt + (size_t)mcap * PF_BS);
    m.SP = m.JM + (size_t)E * P + ((E * P) & 1);
    m.Pn = (int8_t*)(m.SP + ((n + 1) & ~1));
    m.Mb = m.Pn + (size_t)(n - 1) * PF_BS;
    m.Mq = m.Mb + (size_t)mcap * PF_BS;
    m.Bp = m.Mq + (size_t)mcap * PF_BS;
    return m;
}
__device__ __forceinline__ void load_model_mp(const KArgs& A, SmemMP& m) {
    const int EP = A.E * A.P;
    for (int i = threadIdx.x; i < EP; i += blockDim.x) { m.I2[i] = A.inv2Np[i]; m.MT[i] = A.mig_tot[i]; m.JM[i] = A.join_map[i]; m.CI[i] = A.cum_coal[i]; m.CM[i] = A.cum_mig[i]; }
    for (int i = threadIdx.x; i < A.E; i += blockDim.x) m.TJ[i] = A.next_join[i];
    for (int i = threadIdx.x; i < EP * A.P; i += blockDim.x) m.MR[i] = A.mig_rate[i];
    for (int i = threadIdx.x; i < A.n; i += blockDim.x) m.SP[i] = A.sample_pop[i];
}
__device__ __forceinline__ MLane make_mlane(const KArgs& A, SmemMP& m) {
    MLane ml;
    ml.Pn = m.Pn + threadIdx.x; ml.Mt = m.Mt + threadIdx.x; ml.Mb = m.Mb + threadIdx.x; ml.Mq = m.Mq + threadIdx.x; ml.Bp = m.Bp + threadIdx.x;
    ml.nm = 0; ml.P = A.P; ml.mcap = A.mcap;
    ml.I2 = m.I2; ml.MR = m.MR; ml.MT = m.MT; ml.CI = m.CI; ml.CM = m.CM; ml.TJ = m.TJ; ml.JM = m.JM; ml.SP = m.SP; ml.vbm = A.vb_mig;
    ml.err = 0;
    return ml;
}
__device__ __forceinline__ void mp_report(const KArgs& A, const MLane& ml) {
    // an error of an earlier row stays the one reported (what follows it on a broken state is a consequence)
    if (ml.err && !A.ctrl->err) A.ctrl->err = ml.err == 1 ? ERR_MIG_OVERFLOW : (ml.err == 2 ? ERR_MP_INTERNAL : ERR_NO_COALESCENCE);
}
__device__ __forceinline__ double piece_ref(unsigned pstart, unsigned npieces) {
    return __longlong_as_double((long long)((unsigned long long)pstart | ((unsigned long long)npieces << 32)));
}
__device__ __forceinline__ void store_mp_state(const KArgs& A, const DState& st, const Lane& ln, const MLane& ml, long long p) {
    const int n = A.n;
    for (int r = 0; r < n - 1; ++r) st.Pn[(size_t)r * A.Np + p] = LPn(ml, r);
    st.nm[p] = ml.nm;
    for (int m = 0; m < ml.nm; ++m) {
        st.Mt[(size_t)m * A.Np + p] = LMt(ml, m);
        st.Mb[(size_t)m * A.Np + p] = LMb(ml, m);
        st.Mq[(size_t)m * A.Np + p] = LMq(ml, m);
    }
}

__global__ __launch_bounds__(PF_BS) void k_init_mp(KArgs A, double initial_position) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    SmemMP mm = carve_mp(smem, A.n, A.E, A.P, A.mcap);
    load_model(A, m);
    load_model_mp(A, mm);
    __syncthreads();
    long long p = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (p == 0) {
        Ctrl* c = A.ctrl;
        c->cur_pos = initial_position;
        c->logl = 0; c->inv_T = 1; c->T = 1; c->flag = 0; c->cur = 0; c->gen = 0; c->n_resample = 0; c->lver = 0;
        c->first_epoch = A.E; c->err = 0; c->delayed_opp = 0; c->delayed_count = 0; c->count_active = 0; c->end_seq = 0;
        c->g_retain = 0; c->g_safe = 0; c->delay_peak = 0; c->n_delay_evict = 0; c->pending_fin = 0;
        for (int k = 0; k < PF_RING; ++k) c->ri[k].g_retain = 0;      // (what Ctrl::g_safe is read from in the first rows of a sweep) c->nbx_used = A.nbx; c->gen_prev = 0; c->nres_prev = 0;
        for (int e = 0; e < A.E; ++e) { c->counted_to[e] = 0; c->update_to[e] = 0; c->g_lo[e] = 0; c->g_hi[e] = 0; }
        A.gen_x0[0] = 0.0;
    }
    if (p >= A.Np) return;
    const int n = A.n;
    Lane ln = make_lane(A, m, p);
    MLane ml = make_mlane(A, mm);
    ln.ebuf = -dlog(uni(ln));
    unsigned widx = 0;
    PLog pl;
    pl.base = A.plog + (size_t)p * A.pcap * 3; pl.cap = A.pcap; pl.idx = 0; pl.pos = 0; pl.on = true; pl.fopen = false; pl.ropen = false;
    // every coalescence of the initial tree is logged as a type-2 record at position 0 (particle.cpp:251-300)
    double w0 = 1.0 / (double)A.Np;
    mp_build_initial_tree<true>(ln, ml, pl, [&](int i, unsigned p0, unsigned np_, double tc, unsigned below) {
        if (ln.vbc) { w0 *= ln.upd_fac; ln.upd_fac = 1.0; }
        double* rec = rec_ptr(A, p, widx);
        rec[0] = 0.0; rec[1] = 0.0; rec[2] = 0.0;
        rec[3] = piece_ref(p0, np_);
        rec[4] = __longlong_as_double((long long)make_meta(2, A.E - 1, A.E - 1, i, 0, below));
        for (int r = 0; r < n - 1; ++r) rec[5 + r] = 0.0;
        (void)tc;
        ++widx;
    }, A.rec_trees ? m.t0 + threadIdx.x : nullptr);
    mp_report(A, ml);
    double nb = sample_next_base_guided(ln, 0.0, A.g_K, A.g_pos, A.g_rho, 0);
    const DState st = A.st0;
    for (int r = 0; r < n - 1; ++r) {
        st.S[(size_t)r * A.Np + p] = LS(ln, r);
        st.C[(size_t)(2 * r) * A.Np + p] = LC(ln, r, 0);
        st.C[(size_t)(2 * r + 1) * A.Np + p] = LC(ln, r, 1);
    }
    store_mp_state(A, st, ln, ml, p);
    st.w_post[p] = w0;
    st.w_pilot[p] = w0;
    st.next_base[p] = nb;
    st.x_mark[p] = 0.0;
    st.Ltree[p] = ln.Ltree;
    st.mark_limit[p] = A.E - 1;
    if (A.n_bias > 0 || A.g_K > 0) { A.st0.total_delayed[p] = 1.0; A.st0.dcount[p] = 0; }
    if (A.g_K > 0) A.st0.ridx[p] = 0;
    A.rng_ctr[p] = ln.ctr;
    A.ebuf[p] = ln.ebuf;
    A.widx[p] = widx;
    A.pidx[p] = pl.idx;
    A.gstart[p] = 0;
}

// BIASED = false: no focused sampling and no guide; their code is compiled out
template <bool BIASED>
__global__ __launch_bounds__(PF_BS) void k_extend_mp(KArgs A, long long s) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    SmemMP mm = carve_mp(smem, A.n, A.E, A.P, A.mcap);
    load_model(A, m);
    load_model_mp(A, mm);
    __shared__ double sBH[PF_BIAS_MAX + 2], sBS[PF_BIAS_MAX + 1];      // focused sampling: band boundaries / strengths
    if (threadIdx.x < PF_BIAS_MAX + 2) {
        sBH[threadIdx.x] = A.bias_H[threadIdx.x];
        if (threadIdx.x < PF_BIAS_MAX + 1) sBS[threadIdx.x] = A.bias_S[threadIdx.x];
    }
    __syncthreads();
    const Ctrl* c = A.ctrl;
    const int n = A.n;
    const int cur = __builtin_amdgcn_readfirstlane(c->cur);
    const long long p = (long long)blockIdx.x * PF_BS + threadIdx.x;
    const bool active = p < A.Np;
    const int lane = threadIdx.x & 63;
    double w_post = 0.0, w_pilot = 0.0;
    MP_TICK(tk_begin);
#ifdef PF_STAMPS
    if (threadIdx.x < PF_STAMP_W) g_mp_acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long* stamp_out = (A.stamps && s < A.stamp_rows) ? A.stamps + ((size_t)s * A.nc + (size_t)(p >> 6)) * PF_STAMP_W : nullptr;
#endif
    const bool guided = BIASED && A.g_K > 0;
    const bool biased = BIASED && (A.n_bias > 0 || guided);          // a guide alone runs with one band of strength 1
    bool has_pending = false;
    if (active) {
        const DState st = state_slot(A, cur);
        Lane ln = make_lane(A, m, p);
        MLane ml = make_mlane(A, mm);
        DStore ds;
        d_bind(ds, A, st, p);
        ds.count = 0; ds.total = 1.0;
        if (biased) { ds.count = st.dcount[p]; ds.total = st.total_delayed[p]; }
        int ridx = guided ? st.ridx[p] : 0;
        for (int r = 0; r < n - 1; ++r) {
            LS(ln, r) = st.S[(size_t)r * A.Np + p];
            LC(ln, r, 0) = st.C[(size_t)(2 * r) * A.Np + p];
            LC(ln, r, 1) = st.C[(size_t)(2 * r + 1) * A.Np + p];
            LPn(ml, r) = st.Pn[(size_t)r * A.Np + p];
        }
        ml.nm = st.nm[p];
        for (int q = 0; q < ml.nm; ++q) {
            LMt(ml, q) = st.Mt[(size_t)q * A.Np + p];
            LMb(ml, q) = st.Mb[(size_t)q * A.Np + p];
            LMq(ml, q) = st.Mq[(size_t)q * A.Np + p];
        }
        w_post = st.w_post[p];
        w_pilot = st.w_pilot[p];
        double next_base = st.next_base[p];
        double x_mark = st.x_mark[p];
        int mark_limit = st.mark_limit[p];
        ln.Ltree = st.Ltree[p];
        ln.ctr = A.rng_ctr[p];
        ln.ebuf = A.ebuf[p];
        unsigned widx = A.widx[p];
        PLog pl;
        pl.base = A.plog + (size_t)p * A.pcap * 3; pl.cap = A.pcap; pl.idx = A.pidx[p]; pl.pos = pl.idx % pl.cap; pl.on = true;
        pl.fopen = false; pl.ropen = false;
        double* tmp0 = m.t0 + threadIdx.x;
        double* tmp1 = m.t1 + threadIdx.x;

        const int8_t* data = A.seg_alleles + (size_t)s * n;
        const double seg_end = A.seg_start[s] + A.seg_len[s];
        const double extend_to = seg_end < A.L ? seg_end : A.L;
        const int limit = A.seg_limit[s];
        int missing = 0;
        for (int i = 0; i < n; ++i) missing += data[i] == -1;
        int leaf_status = 0;
        if (missing == 0) leaf_status = 1;
        if (missing == n) leaf_status = -1;

        double updated_to = c->cur_pos;
        double B;
        if (leaf_status == -1) B = 0;
        else if (leaf_status == 1) B = ln.Ltree;
        else B = tracked_len_lane(ln, data, tmp0);

        MP_TICK(tk_loaded);
        MP_ACC(ml, 0, tk_begin, tk_loaded);
        while (updated_to < extend_to) {
            MP_ACC(ml, 14, 0, 1);
            MP_TICK(tu0);
            double new_to = extend_to < next_base ? extend_to : next_base;
            double f = fastexp(-A.mu * B * (new_to - updated_to));
            w_post *= f;
            w_pilot *= f;
            if (guided) {
                // importance_weight_over_segment (particle.cpp:1138-1181)
                const double dist = new_to - updated_to;
                const double target_rate = dist * A.rho * ln.Ltree;
                const double sampled_rate = dist * A.g_rho[ridx] * ln.Ltree;
                const double iws = fastexp(sampled_rate - target_rate);
                w_post *= iws;
                w_pilot *= iws;
            }
            updated_to = new_to;
            if (guided && updated_to < extend_to && ridx + 1 < A.g_K && updated_to == A.g_pos[ridx + 1]) {
                // reached a change of the guide rate: no genealogy change, new draw under the new rate
                ridx += 1;
                next_base = sample_next_base_guided(ln, updated_to, A.g_K, A.g_pos, A.g_rho, ridx);
                continue;
            }
            if (updated_to < extend_to) {
                double* rec = rec_ptr(A, p, widx);
                rec[0] = x_mark;
                rec[1] = updated_to;
                for (int r = 0; r < n - 1; ++r) rec[5 + r] = LS(ln, r);
                int rp = 0, sb = 0;
                double h, tc, sp_removed;
                bool changed;
                double iw = 1.0, rbiw = 1.0;
                if (guided)
                    sample_point_guided(ln, sBH, sBS, A.n_bias + 1, A.g_leaf + (size_t)ridx * n, A.rho / A.g_rho[ridx], &rp, &sb, &h,
                                        &iw, &rbiw, tmp0);
                else if (biased) { sample_point_biased(ln, sBH, sBS, A.n_bias + 1, &rp, &sb, &h, &iw); rbiw = iw; }
                else sample_point(ln, &rp, &sb, &h);
                const unsigned desc = A.lmap_opp ? lane_desc_mask(ln, LC(ln, rp, sb), tmp0) : 0u;
                unsigned p0 = pl.idx;
                double tfirst = 0.0;
                MP_TICK(tu1);
                MP_ACC(ml, 8, tu0, tu1);
                unsigned span = 0;
                mp_genealogy_rest<true>(ln, ml, pl, limit, rp, sb, h, &tc, &sp_removed, &changed, &tfirst, &span);
                if (ln.vbc) { w_post *= ln.upd_fac; w_pilot *= ln.upd_fac; ln.upd_fac = 1.0; }
                rec[2] = h;
                rec[3] = piece_ref(p0, pl.idx - p0);
                rec[4] = __longlong_as_double((long long)make_meta(0, mark_limit, limit, n, desc, span));
                ++widx;
                if (ml.err) break;
                MP_TICK(tu2);
                if (leaf_status == 0) B = tracked_len_lane(ln, data, tmp0);
                if (leaf_status == 1) B = ln.Ltree;
                if (biased) {
                    // particle.cpp:866-891: immediate vs delayed application of the importance weight
                    const int nbands = A.n_bias + 1;
                    const double delay_height = (A.delay_type & 3) == 0 ? h : ((A.delay_type & 3) == 2 ? tfirst : tc);
                    int idx = 0;
                    while (idx + 1 < nbands + 1 && sBH[idx + 1] < delay_height) ++idx;
                    if (idx >= nbands) idx = nbands - 1;
                    if (sBS[idx] == 1.0 && !(A.delay_type & 4)) { w_post *= rbiw; w_pilot *= rbiw; iw /= rbiw; }   // bit 2: every factor delayed (pf_model.delay_type)
                    const double delay = A.app_delays[epoch_of(ln, delay_height)];
                    d_adjust_with_delay(ds, w_post, w_pilot, iw, delay, updated_to);
                }
                next_base = sample_next_base_guided(ln, updated_to, A.g_K, A.g_pos, A.g_rho, ridx);
                x_mark = updated_to;
                mark_limit = limit;
                MP_TICK(tu3);
                MP_ACC(ml, 9, tu2, tu3);
            }
        }
        MP_TICK(tk_loop);
        mp_report(A, ml);
        if (biased) {
            // apply the factors that fell due during this extension (particle.cpp:910-916)
            for (;;) {
                if (ds.count == 0) break;
                double pm = ds.pos[0];
                for (int i = 1; i < ds.count; ++i) { double pi = ds.pos[(size_t)i * ds.Np]; if (pi < pm) pm = pi; }
                if (!(pm < extend_to)) break;
                d_apply_earliest(ds, w_pilot);
            }
            st.dcount[p] = ds.count;
            st.total_delayed[p] = ds.total;
            if (guided) st.ridx[p] = ridx;
            has_pending = ds.count > 0;
        }

        if (A.seg_state[s] == 0) {
            // update_weight_at_site: marginalise over phasings of unphased hets (pc.cpp:138-224)
            const bool dephase = A.flags & 2;
            const bool anc = A.flags & 1;
            unsigned one_mask = 0, zero_mask = 0, het_pairs = 0;
            int ncfg = 1;
            for (int i = 0; i < n; ++i) {
                if (data[i] == 1) one_mask |= 1u << i;
                if (data[i] == 0) zero_mask |= 1u << i;
            }
            for (int i = 0; i + 1 < n; i += 2) {
                bool het = (data[i] == 2) || (dephase && data[i] + data[i + 1] == 1);
                if (het) {
                    ncfg *= 2;
                    het_pairs |= 1u << i;
                    one_mask &= ~(3u << i); zero_mask &= ~(3u << i);
                    zero_mask |= 1u << i;
                    one_mask |= 1u << (i + 1);
                }
            }
            double norm = 1.0 / (double)ncfg;
            double lik = 0;
            for (;;) {
                lik += site_lik_lane(ln, one_mask, zero_mask, anc, tmp0, tmp1);
                if (ncfg == 1) break;
                bool more = false;
                for (int i = 0; i + 1 < n; i += 2) {
                    if (!((het_pairs >> i) & 1)) continue;
                    if ((zero_mask >> i) & 1) {
                        zero_mask &= ~(1u << i); one_mask |= 1u << i;
                        one_mask &= ~(1u << (i + 1)); zero_mask |= 1u << (i + 1);
                        more = true;
                        break;
                    }
                    one_mask &= ~(1u << i); zero_mask |= 1u << i;
                    zero_mask &= ~(1u << (i + 1)); one_mask |= 1u << (i + 1);
                }
                if (!more) break;
            }
            lik *= norm;
            w_post *= lik;
            w_pilot *= lik;
        }
        MP_TICK(tk_lik);
        MP_ACC(ml, 10, tk_loop, tk_lik);

        for (int r = 0; r < n - 1; ++r) {
            st.S[(size_t)r * A.Np + p] = LS(ln, r);
            st.C[(size_t)(2 * r) * A.Np + p] = LC(ln, r, 0);
            st.C[(size_t)(2 * r + 1) * A.Np + p] = LC(ln, r, 1);
        }
        store_mp_state(A, st, ln, ml, p);
        st.w_post[p] = w_post;
        st.w_pilot[p] = w_pilot;
        st.next_base[p] = next_base;
        st.x_mark[p] = x_mark;
        st.mark_limit[p] = mark_limit;
        st.Ltree[p] = ln.Ltree;
        A.rng_ctr[p] = ln.ctr;
        A.ebuf[p] = ln.ebuf;
        A.widx[p] = widx;
        A.pidx[p] = pl.idx;
        for (int r = 0; r < n - 1; ++r) A.snap_S[A.sp][(size_t)r * A.Np + p] = LS(ln, r);
        A.snap_w[A.sp][p] = w_post; A.snap_xm[A.sp][p] = x_mark; A.snap_ml[A.sp][p] = mark_limit; A.snap_widx[A.sp][p] = widx;
        MP_TICK(tk_stored);
        MP_ACC(ml, 11, tk_lik, tk_stored);
        MP_ACC(ml, 15, tk_begin, tk_stored);
    }
#ifdef PF_STAMPS
    __syncthreads();
    if (stamp_out && threadIdx.x < PF_STAMP_W) stamp_out[threadIdx.x] = g_mp_acc[threadIdx.x];
#endif
    double sp = wave_tree_sum(w_post);
    double sq = wave_tree_sum(w_pilot * w_pilot);
    double sc = wave_hs_scan(w_pilot, lane);
    double scp = wave_hs_scan(w_post, lane);
    double scm = wave_max_scan_d(sc, lane);
    long long chunk = p >> 6;
    if (active) { A.scan1[p] = sc; A.scanp2[A.sp][p] = scp; A.scan1m[p] = scm; }
    if (lane == 63 && chunk < (A.Np + 63) / 64) {
        A.chunk_post[chunk] = sp;
        A.chunk_sq[chunk] = sq;
        A.chunk_pil[chunk] = sc;
        A.chunk_pp[chunk] = scp;
        A.chunk_mx1[chunk] = scm;
    }
    if (biased) {
        unsigned long long pend = __ballot(has_pending);
        if (lane == 0 && chunk < (A.Np + 63) / 64) A.chunk_dpend[chunk] = __popcll(pend);
    }
}

// ------------------------------------------------------------------ structured models, register-resident tree (n <= 8)
// The row kernel of the structured filter for small samples: the update of pf_mp_reg.h, state read from and written to
// the same arrays as k_extend_mp (the two are interchangeable row by row; tests run both against the oracle).
// LDS: the epoch tables of the model and, per lane, the migration events of its tree.
// particles per wavefront of the register-tree row kernel (see k_extend_mpr): the fewer, the less a wavefront serialises
// the divergent paths of its lanes -- as long as all workgroups of a launch are resident at once.  Without focused sampling
// the kernel needs 223 VGPRs, two wavefronts share a SIMD, and four quarter-filled wavefronts per workgroup fit (C5 shape:
// 7.64e3 segments/s against 7.38e3 with two half-filled and 7.30e3 with one full wavefront, alternating runs on one box);
// the focused-sampling instantiation needs more than 256 registers (one wavefront per SIMD) and stays at two.
#ifndef PF_MPR_LANES_PLAIN
#define PF_MPR_LANES_PLAIN 16
#endif
#ifndef PF_MPR_LANES_BIASED
#define PF_MPR_LANES_BIASED 32
#endif
struct SmemMPR { double* T; double* TJ; double* I2; double* MT; double* CI; double* CM; double* MR; double* Mt; int* JM; int* EJ; int8_t* Mb; int8_t* Mq; };
__host__ __device__ static size_t smem_mpr_bytes(int E, int P, int mcap) {
    const size_t EP = (size_t)E * P;
    return ((size_t)2 * PF_EPAD + 4 * EP + EP * P + (size_t)mcap * PF_BS) * 8 + (((EP + 1) & ~(size_t)1) + PF_EPAD) * 4 + (size_t)2 * mcap * PF_BS;
}
__device__ __forceinline__ SmemMPR carve_mpr(double* base, int E, int P, int mcap) {
    const size_t EP = (size_t)E * P;
    SmemMPR m;
    m.T = base; m.TJ = m.T + PF_EPAD; m.I2 = m.TJ + PF_EPAD; m.MT = m.I2 + EP; m.CI = m.MT + EP; m.CM = m.CI + EP;
    m.MR = m.CM + EP; m.Mt = m.MR + EP * P;
    m.JM = (int*)(m.Mt + (size_t)mcap * PF_BS);
    m.EJ = m.JM + ((EP + 1) & ~(size_t)1);
    m.Mb = (int8_t*)(m.EJ + PF_EPAD);
    m.Mq = m.Mb + (size_t)mcap * PF_BS;
    return m;
}

// LA = lanes of a wavefront that carry a particle.  A workgroup still owns 64 consecutive particles (the unit of the
// canonical reductions), spread over 64 / LA wavefronts: a row lasts as long as its slowest wavefront, the lanes of a
// wavefront wait for the one with the most recombinations and the longest walk, and with one workgroup per 64
// particles most of the 1024 SIMDs have nothing to do -- half-filled wavefronts on twice as many SIMDs wait for the
// maximum over 32 lanes instead of 64 and serialise fewer divergent paths.  The weights meet in LDS, and the first
// wavefront does the reductions over the 64 particles exactly as the full wavefront did.
// TREES (-arg): every record carries the samples below the cut branch and below the node the update creates, and
// neither the record ring nor the piece ring may wrap.
// PIPE: the extend role of the row pipeline for structured models (k_sweep_xmp): the workgroup takes the decision on the
// previous row itself (decide_row), its first wavefront works out the offspring offsets and finds the parents of the
// workgroup's 64 slots by search (the code of extend_reg_body's prologue, one particle per lane of that wavefront), the
// previous row is read from one slot of the state ring and this row written to the next -- nothing from k_decide is read.
template <int NM, bool BIASED, int LA, bool TREES, bool PIPE, class KA>
__device__ __forceinline__ void extend_mpr_body(const KA& A, long long s, int fuse, PipeRow PR = PipeRow()) {
    constexpr int NI = RTree<NM>::NI;
    constexpr int BS = 64 * (64 / LA);
    MP_TICK(tk_begin);                      // profiling builds: "load state" counts from here (tables, completion of the previous row)
    extern __shared__ double smem[];
    SmemMPR mm = carve_mpr(smem, A.E, A.P, A.mcap);
    {
        const int EP = A.E * A.P;
        for (int i = threadIdx.x; i < PF_EPAD; i += blockDim.x) {
            mm.T[i] = i < A.E ? A.T[i] : PF_INF;              // padded: the four-way search needs no bounds checks
            mm.TJ[i] = i < A.E ? A.next_join[i] : PF_INF;
            mm.EJ[i] = i < A.E ? A.next_join_epoch[i] : A.E;
        }
        for (int i = threadIdx.x; i < EP; i += blockDim.x) {
            mm.I2[i] = A.inv2Np[i]; mm.MT[i] = A.mig_tot[i]; mm.CI[i] = A.cum_coal[i]; mm.CM[i] = A.cum_mig[i]; mm.JM[i] = A.join_map[i];
        }
        for (int i = threadIdx.x; i < EP * A.P; i += blockDim.x) mm.MR[i] = A.mig_rate[i];
    }
    __shared__ double sBH[PF_BIAS_MAX + 2], sBS[PF_BIAS_MAX + 1];      // focused sampling: band boundaries / strengths
    __shared__ double sWpost[64], sWpilot[64];
    __shared__ int sPend[64];
    if (threadIdx.x < PF_BIAS_MAX + 2) {
        sBH[threadIdx.x] = A.bias_H[threadIdx.x];
        if (threadIdx.x < PF_BIAS_MAX + 1) sBS[threadIdx.x] = A.bias_S[threadIdx.x];
    }
    const Ctrl* c = A.ctrl;
    // PIPE: what the decision on the previous row reads is requested before the barrier below waits for the tables, so that
    // the two round trips to memory overlap
    RowPre pre_row;
    long long pre_nres = 0; int pre_gen = 0;
    // ... and so is what the parent search of a resampling row reads: the pilot scan of the workgroup's own slot and of the
    // wavefronts around it (offspring stay close to their parents' slots; a range outside is staged the old way)
    double pre_sm = 0.0, spec_sm[PF_PIPE_STAGE * 64 / BS];
    int spec_lo = 0;
    if constexpr (PIPE) {
        if (PR.complete != 0) {
            const int fs0 = __builtin_amdgcn_readfirstlane(PR.slot_prev >= 0 ? PR.slot_prev : c->cur);
            pre_row = row_preload<BS>(A, fs0);
            const double* sm0 = A.rg_scan1m + (size_t)fs0 * A.Np;
            spec_lo = (int)blockIdx.x - (PF_PIPE_STAGE - 1) / 2;
            if (spec_lo > A.nc - PF_PIPE_STAGE) spec_lo = A.nc - PF_PIPE_STAGE;
            if (spec_lo < 0) spec_lo = 0;
#pragma unroll
            for (int k = 0; k < PF_PIPE_STAGE * 64 / BS; ++k) {
                const long long src = (long long)spec_lo * 64 + k * BS + threadIdx.x;
                spec_sm[k] = src < A.Np ? sm0[src] : PF_INF;
            }
            {
                const long long qp0 = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
                if (threadIdx.x < 64 && qp0 < A.Np) pre_sm = sm0[qp0];
            }
            pre_nres = c->xr[(fs0 + PF_RING - 1) & (PF_RING - 1)].n_res + c->xr[(fs0 + PF_RING - 1) & (PF_RING - 1)].flag;
            pre_gen = c->xr[(fs0 + PF_RING - 1) & (PF_RING - 1)].gen + c->xr[(fs0 + PF_RING - 1) & (PF_RING - 1)].flag;
        }
    }
    __syncthreads();
    MP_TICK(tk_tables);
    const int n = A.n;
    const int lane = threadIdx.x & 63;
    const int cslot = (int)(threadIdx.x >> 6) * LA + lane;             // place of this lane's particle in the workgroup's 64
    const long long p = (long long)blockIdx.x * 64 + cslot;
    const bool active = lane < LA && p < A.Np;
    // fuse: the previous row was decided (k_decide) but not completed -- this kernel does k_resample's part while it
    // loads: normalisation, or the copy from the parent with the closing record of the old slot (pc.cpp:335-368)
    int cur, from_slot;
    bool completing, gather;
    double inv_prev = 1.0, S1_prev = 0.0, pos_prev = 0.0;
    int G_end = 0, ev_idx = 0;
    long long a_par = p;
    int lo_p = 0, lo_p1 = 0;
    bool first_copy = true;
    unsigned long long tk_decided = 0;
    if constexpr (PIPE) {
        __shared__ long long sPar[64];
        __shared__ int sLoP[64], sLoP1[64], sFirst[64], sRange[2];
        // the decision's tables live where the migration events of the lanes go afterwards (they are loaded once the previous row
        // is completed): with an area of their own the workgroup would need 85 KB and a CU would hold one instead of two
        const bool shared_area = pipe_lds_doubles(A.nc) <= (size_t)A.mcap * PF_BS;          // else behind everything (a small mig_cap)
        PipeLds q = pipe_carve(shared_area ? mm.Mt : (double*)((char*)smem + smem_mpr_bytes(A.E, A.P, A.mcap)), A.nc);
        const int fs = __builtin_amdgcn_readfirstlane(PR.slot_prev >= 0 ? PR.slot_prev : c->cur);
        cur = PR.slot_out; from_slot = fs;
        completing = PR.complete != 0;
        gather = false;
        pos_prev = PR.pos_prev;
        if (completing) {
            const int row_slot = fs;
            const RowPre pre = pre_row;
            // the row before it: what the extend role of the previous launch noted (the bookkeeping role runs on another
            // stream here and is not waited for)
            const long long n_res = pre_nres;
            G_end = pre_gen;
            ev_idx = (int)n_res;
            RowDecision d = decide_row<true, BS>(A, q, row_slot, n_res, pre);
            tk_decided = wall_clock64();
            inv_prev = d.inv; S1_prev = d.S1;
            gather = d.flag != 0;
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                Ctrl* cw = A.ctrl;
                cw->xr[row_slot].n_res = n_res; cw->xr[row_slot].gen = G_end; cw->xr[row_slot].flag = d.flag;
            }
            if (gather) {
                const double dn = (double)A.Np;
                const double invS1 = 1.0 / d.S1;
                const double* sm = A.rg_scan1m + (size_t)row_slot * A.Np;
                const long long qp = (long long)blockIdx.x * 64 + lane;          // first wavefront: one of the workgroup's slots per lane
                const bool qa = threadIdx.x < 64 && qp < A.Np;
                const int ch_own = (int)blockIdx.x;
                const double lhs = ((double)qp + d.u) * d.S1;
                int pch = 0, lo_next = 0;
                if (threadIdx.x < 64) {
                    if (qa) {
                        const double w = pipe_chunk_offset(q, ch_own) + pre_sm;
                        const double v_own = q.pmx[ch_own] > w ? q.pmx[ch_own] : w;      // largest pilot prefix sum up to slot qp
                        lo_next = qp + 1 < A.Np ? pipe_lo_from(v_own, dn, A.Np, d.S1, invS1, d.u) : (int)A.Np;
                        int lo_c = 0, hi_c = A.nc - 1;
                        while (lo_c < hi_c) {
                            int mid = (lo_c + hi_c) >> 1;
                            if (lhs < dn * q.pmx[mid + 1]) hi_c = mid; else lo_c = mid + 1;
                        }
                        pch = lo_c;
                    }
                    int cmin = qa ? pch : 0x7fffffff, cmax = qa ? pch : -1;
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) {
                        int o1 = __shfl_xor(cmin, m, 64), o2 = __shfl_xor(cmax, m, 64);
                        cmin = o1 < cmin ? o1 : cmin; cmax = o2 > cmax ? o2 : cmax;
                    }
                    if (lane == 0) { sRange[0] = cmin; sRange[1] = cmax; }
                }
                __syncthreads();
                int cmin = sRange[0];
                const int cmax = sRange[1];
                int nst = cmax - cmin + 1;
                if (nst > PF_PIPE_STAGE) nst = PF_PIPE_STAGE;
                if (nst < 0) nst = 0;
                if (cmin >= spec_lo && cmax < spec_lo + PF_PIPE_STAGE) {
                    cmin = spec_lo; nst = PF_PIPE_STAGE;           // the scans requested with the prologue cover the range
#pragma unroll
                    for (int k = 0; k < PF_PIPE_STAGE * 64 / BS; ++k) q.stage[k * BS + threadIdx.x] = spec_sm[k];
                } else {
                    for (int idx = threadIdx.x; idx < nst * 64; idx += BS) {
                        const long long src = (long long)cmin * 64 + idx;
                        q.stage[idx] = src < A.Np ? sm[src] : PF_INF;
                    }
                }
                __syncthreads();
                if (threadIdx.x < 64) {
                    int lp = __shfl_up(lo_next, 1, 64);
                    if (lane == 0) lp = qp > 0 ? pipe_lo_from(q.pmx[ch_own], dn, A.Np, d.S1, invS1, d.u) : 0;
                    const unsigned long long bal = __ballot(qa && lo_next > lp);
                    if (lane == 0) A.rg_blkcnt[(size_t)row_slot * ((A.Np + 255) / 256) * A.blk_gran + blockIdx.x] = __popcll(bal);
                    if (qa) {
                        const double coff_p = pipe_chunk_offset(q, pch);
                        const double pm_p = q.pmx[pch];
                        const bool staged = pch - cmin < nst;
                        auto val_at = [&](int l) -> double {           // largest prefix sum up to particle pch*64 + l
                            long long idx = (long long)pch * 64 + l;
                            double smv = staged ? q.stage[(pch - cmin) * 64 + l] : (idx < A.Np ? sm[idx] : PF_INF);
                            double w = coff_p + smv;
                            return pm_p > w ? pm_p : w;
                        };
                        int lo_l = 0, hi_l = 63;
                        while (lo_l < hi_l) {
                            int mid = (lo_l + hi_l) >> 1;
                            if (lhs < dn * val_at(mid)) hi_l = mid; else lo_l = mid + 1;
                        }
                        long long a = (long long)pch * 64 + lo_l;
                        if (a > A.Np - 1) a = A.Np - 1;
                        bool fc;
                        const int la = (int)(a & 63);
                        if (qp == 0 || a == 0) fc = (qp == 0);
                        else {
                            double vprev = la > 0 ? val_at(la - 1) : pm_p;      // largest prefix sum up to a - 1
                            if (a != (long long)pch * 64 + lo_l) {             // clamped: recompute on the true chunk of a - 1
                                long long am = a - 1;
                                int chm = (int)(am >> 6);
                                double w = pipe_chunk_offset(q, chm) + sm[am];
                                vprev = q.pmx[chm] > w ? q.pmx[chm] : w;
                            }
                            fc = ((((double)(qp - 1)) + d.u) * d.S1 < dn * vprev);
                        }
                        sPar[lane] = a; sFirst[lane] = fc ? 1 : 0; sLoP[lane] = lp; sLoP1[lane] = lo_next;
                        // the offspring table of the generation that ends here, for the ledger and -arg
                        int* lo_tab = A.lo + (size_t)(G_end % A.Gcap) * (A.Np + 1);
                        lo_tab[qp] = lp;
                        if (qp == A.Np - 1) lo_tab[A.Np] = (int)A.Np;
                    }
                }
                __syncthreads();
                if (active) { a_par = sPar[cslot]; first_copy = sFirst[cslot] != 0; lo_p = sLoP[cslot]; lo_p1 = sLoP1[cslot]; }
            }
        }
        __syncthreads();          // the tables are dead from here on: the lanes' event lists take their place
    } else {
        cur = __builtin_amdgcn_readfirstlane(c->cur);
        completing = fuse != 0;
        gather = completing && __builtin_amdgcn_readfirstlane(c->flag) != 0;
        from_slot = gather ? (cur ^ 1) : cur;
    }
    double w_post = 0.0, w_pilot = 0.0;
    MP_TICK(tk_searched);
#ifdef PF_STAMPS
    if (threadIdx.x < PF_STAMP_W) g_mp_acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long* stamp_out = (A.stamps && s < A.stamp_rows) ? A.stamps + ((size_t)s * A.nc + (size_t)(p >> 6)) * PF_STAMP_W : nullptr;
#endif
    const bool guided = BIASED && A.g_K > 0;
    const bool biased = BIASED && (A.n_bias > 0 || guided);          // a guide alone runs with one band of strength 1
    bool has_pending = false;
    if (active) {
        const DState st = state_slot(A, cur);
        const DState from = state_slot(A, from_slot);
        const long long a = PIPE ? (gather ? a_par : p) : (gather ? (long long)A.parent[p] : p);
        if (!PIPE && completing && p == 0) { Ctrl* cw = A.ctrl; cw->gen_prev = c->gen; cw->nres_prev = c->n_resample; }
        RTree<NM> t;
        MRLane ml;
        ml.Mt = mm.Mt + cslot; ml.Mb = mm.Mb + cslot; ml.Mq = mm.Mq + cslot;
        ml.P = A.P; ml.mcap = A.mcap; ml.I2 = mm.I2; ml.MR = mm.MR; ml.MT = mm.MT; ml.CI = mm.CI; ml.CM = mm.CM; ml.TJ = mm.TJ; ml.JM = mm.JM; ml.EJ = mm.EJ;
        ml.vbm = A.vb_mig; ml.err = 0; ml.pn = 0; ml.bp = 0; ml.sp = 0;
        for (int i = 0; i < n; ++i) ml.sp |= (unsigned)A.sample_pop[i] << (2 * i);
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            t.S[r] = 0.0; t.C0[r] = 0; t.C1[r] = 0;
            if (r < n - 1) {
                t.S[r] = from.S[(size_t)r * A.Np + a];
                t.C0[r] = from.C[(size_t)(2 * r) * A.Np + a];
                t.C1[r] = from.C[(size_t)(2 * r + 1) * A.Np + a];
                ml.pn |= (unsigned)from.Pn[(size_t)r * A.Np + a] << (2 * r);
            }
        }
        // Everything of the particle that does not depend on anything else is requested in ONE round: the tree above, the
        // number of migration events, the first eight events whatever that number is (the arrays hold mcap >= 8 entries; what
        // lies beyond the count is ignored), and the scalars of the state.  Loads placed behind the LDS stores of the event
        // lists cannot be moved above them by the compiler: each group was a memory round trip of its own.
        const int nm_ld = from.nm[a];
        constexpr int FB = 12;                       // events of the first round (a tree carries ten on average)
        double tv0[FB];
        int8_t bv0[FB], qv0[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int q = j < A.mcap ? j : A.mcap - 1;
            tv0[j] = from.Mt[(size_t)q * A.Np + a];
            bv0[j] = from.Mb[(size_t)q * A.Np + a];
            qv0[j] = from.Mq[(size_t)q * A.Np + a];
        }
        const double ld_wpost = from.w_post[a], ld_wpilot = from.w_pilot[a], ld_next = from.next_base[a], ld_xmark = from.x_mark[a];
        const int ld_ml = from.mark_limit[a];
        const double ld_Ltree = from.Ltree[a], ld_ebuf = A.ebuf[p];
        const unsigned long long ld_ctr = A.rng_ctr[p];
        const unsigned ld_widx = (PIPE && completing) ? A.rg_widx[(size_t)from_slot * A.Np + p] : A.widx[p];
        const unsigned ld_pidx = A.pidx[p];
        ml.nm = nm_ld;
#pragma unroll
        for (int j = 0; j < FB; ++j)
            if (j < ml.nm) { LMt(ml, j) = tv0[j]; LMb(ml, j) = bv0[j]; LMq(ml, j) = qv0[j]; }
        // the rest eight at a time, their loads issued before the first LDS store
        for (int q0 = FB; q0 < ml.nm; q0 += 8) {
            double tv[8];
            int8_t bv[8], qv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + j < ml.nm ? q0 + j : ml.nm - 1;
                tv[j] = from.Mt[(size_t)q * A.Np + a];
                bv[j] = from.Mb[(size_t)q * A.Np + a];
                qv[j] = from.Mq[(size_t)q * A.Np + a];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q0 + j < ml.nm) { LMt(ml, q0 + j) = tv[j]; LMb(ml, q0 + j) = bv[j]; LMq(ml, q0 + j) = qv[j]; }
        }
        MP_TICK(tk_lists);
        RCtx cx;
        cx.T = mm.T; cx.I = nullptr; cx.H = nullptr; cx.E = A.E; cx.n = n; cx.L = A.L; cx.mu = A.mu; cx.rho = A.rho;
        cx.seed = A.seed; cx.slot = (unsigned)p; cx.stream = 0; cx.u_nb = 0.0;
        cx.nb = A.n_bias + 1; cx.bH = sBH; cx.bS = sBS; cx.last_iw = 1.0; cx.last_rbiw = 1.0;
        cx.want_desc = false; cx.want_desc_new = false; cx.last_desc = 0; cx.last_desc_new = 0;
        cx.vbc = A.vb_coal; cx.upd_fac = 1.0;
        cx.gK = BIASED ? A.g_K : 0; cx.gpos = A.g_pos; cx.grho = A.g_rho; cx.gleaf = A.g_leaf;
        cx.ridx = guided ? from.ridx[a] : 0; cx.g_rp = 0; cx.g_sb = 0;
        ml.nep = rmp_node_epochs(cx, t);
        DStore ds;
        d_bind(ds, A, st, p);
        ds.count = 0; ds.total = 1.0;
        if (biased) {
            ds.count = from.dcount[a]; ds.total = from.total_delayed[a];
            if (gather || PIPE)   // the copy constructor copies the pending factors (particle.cpp:122-123); the ring moves them every row
                for (int k = 0; k < ds.count; ++k) {
                    st.dpos[(size_t)k * A.Np + p] = from.dpos[(size_t)k * A.Np + a];
                    st.dfac[(size_t)k * A.Np + p] = from.dfac[(size_t)k * A.Np + a];
                    st.ddelta[(size_t)k * A.Np + p] = from.ddelta[(size_t)k * A.Np + a];
                    st.dk[(size_t)k * A.Np + p] = from.dk[(size_t)k * A.Np + a];
                }
        }
        w_post = ld_wpost;
        w_pilot = ld_wpilot;
        double next_base = ld_next;
        double x_mark = ld_xmark;
        int mark_limit = ld_ml;
        cx.Ltree = ld_Ltree;
        cx.ctr = ld_ctr;
        cx.ebuf = ld_ebuf;
        unsigned widx = ld_widx;
        if (completing) {
            const double inv = PIPE ? inv_prev : c->inv_T;
            if (!gather) {
                w_post *= inv;                             // normalize_probability, pc.cpp:435-437
                w_pilot *= inv;
            } else {
                const int G = PIPE ? G_end : c->gen - 1;   // the generation that ended with the previous row
                const double pos = PIPE ? pos_prev : c->cur_pos;
                const int* lo_tab = A.lo + (size_t)((G + A.Gcap) % A.Gcap) * (A.Np + 1);
                // role of the old slot p: close its stretch if it has offspring
                if (PIPE ? (lo_p1 > lo_p) : (lo_tab[p + 1] > lo_tab[p])) {
                    double* rec = rec_ptr(A, p, widx);
                    rec[0] = from.x_mark[p];
                    rec[1] = pos;
                    rec[2] = 0.0; rec[3] = 0.0;
                    rec[4] = __longlong_as_double((long long)make_meta(1, from.mark_limit[p], -1, n));
                    for (int r = 0; r < n - 1; ++r) rec[5 + r] = from.S[(size_t)r * A.Np + p];
                    ++widx;
                }
                A.gstart[(size_t)((G + 1) % A.Gcap) * A.Np + p] = widx;
                // role of the new slot p: weights of the copy (pc.cpp:350-351), fresh position for all but the first
                const int ev = PIPE ? ev_idx : (int)c->n_resample - 1;
                if (ev < A.max_trace_events) A.ev_parents[(size_t)ev * A.Np + p] = (int)a;
                const double wp = w_post * inv;
                const double wq = w_pilot * inv;
                const double sumn = (PIPE ? S1_prev : c->S1) * inv;
                const double adj = sumn / ((double)A.Np * wq);
                w_post = wp * adj;
                w_pilot = wq * adj;
                x_mark = pos;
                if ((PIPE ? !first_copy : (p != lo_tab[a])) && pos < A.L) next_base = r_sample_next_base<false>(cx, pos);     // pc.cpp:357-368
            }
        }
        PLog pl;
        pl.base = A.plog + (size_t)p * A.pcap * 3; pl.cap = A.pcap; pl.idx = ld_pidx; pl.pos = pl.idx % pl.cap; pl.on = true;
        pl.fopen = false; pl.ropen = false;

        const bool do_extend = !PIPE || PR.extend != 0;        // the flush step of a call only completes the last row
        const int8_t* data = A.seg_alleles + (size_t)s * n;
        const double seg_end = do_extend ? A.seg_start[s] + A.seg_len[s] : 0.0;
        const double extend_to = seg_end < A.L ? seg_end : A.L;
        const int limit = do_extend ? A.seg_limit[s] : 0;
        unsigned one_mask = 0, zero_mask = 0, present_mask = 0, two_mask = 0;
        int missing = 0;
        for (int i = 0; i < n && do_extend; ++i) {
            const int d = data[i];
            missing += d == -1;
            if (d == 1) one_mask |= 1u << i;
            if (d == 0) zero_mask |= 1u << i;
            if (d == 2) two_mask |= 1u << i;
            if (d >= 0) present_mask |= 1u << i;
        }
        int leaf_status = 0;
        if (missing == 0) leaf_status = 1;
        if (missing == n) leaf_status = -1;

        double updated_to = (PIPE && completing) ? pos_prev : c->cur_pos;
        if (!do_extend) updated_to = extend_to;
        double B;
        if (leaf_status == -1) B = 0;
        else if (leaf_status == 1) B = cx.Ltree;
        else B = r_tracked_len(t, n, present_mask);

        MP_TICK(tk_loaded);
        MP_ACC(ml, 0, tk_begin, tk_loaded);
        MP_ACC(ml, 21, tk_begin, tk_tables);
        if (PIPE && tk_decided) { MP_ACC(ml, 22, tk_tables, tk_decided); MP_ACC(ml, 23, tk_decided, tk_searched); }
        MP_ACC(ml, 24, tk_searched, tk_lists);
        MP_ACC(ml, 25, tk_lists, tk_loaded);
        while (updated_to < extend_to) {
            MP_ACC(ml, 14, 0, 1);
            MP_TICK(tu0);
            double new_to = extend_to < next_base ? extend_to : next_base;
            double f = fastexp(-A.mu * B * (new_to - updated_to));
            w_post *= f;
            w_pilot *= f;
            if (guided) {
                // importance_weight_over_segment (particle.cpp:1138-1181)
                const double dist = new_to - updated_to;
                const double target_rate = dist * A.rho * cx.Ltree;
                const double sampled_rate = dist * A.g_rho[cx.ridx] * cx.Ltree;
                const double iws = fastexp(sampled_rate - target_rate);
                w_post *= iws;
                w_pilot *= iws;
            }
            updated_to = new_to;
            if (guided && updated_to < extend_to && cx.ridx + 1 < A.g_K && updated_to == A.g_pos[cx.ridx + 1]) {
                // reached a change of the guide rate: no genealogy change, new draw under the new rate
                cx.ridx += 1;
                next_base = r_sample_next_base<false>(cx, updated_to);
                continue;
            }
            if (updated_to < extend_to) {
                double* rec = rec_ptr(A, p, widx);
                rec[0] = x_mark;
                rec[1] = updated_to;
#pragma unroll
                for (int r = 0; r < NI; ++r) if (r < n - 1) rec[5 + r] = t.S[r];
                int rp = 0, sb = 0, lin = 0;
                double h, tc;
                double iw = 1.0, rbiw = 1.0;
                const double u_point = r_uni(cx);
                if (guided) {
                    r_sample_point_guided(cx, t, cx.nb > 1, u_point, &h);
                    rp = cx.g_rp; sb = cx.g_sb; iw = cx.last_iw; rbiw = cx.last_rbiw;
                } else {
                    if (biased) { r_sample_point_biased(cx, t, u_point, &h, &lin); iw = cx.last_iw; rbiw = iw; }
                    else r_sample_point_plain(cx, t, u_point, &h, &lin);
                    r_lineages_at(t, n, n - 1, h, lin, &rp, &sb);
                }
                const unsigned desc = (TREES || A.lmap_opp) ? r_desc_mask(t, n, rp, sb) : 0u;
                unsigned desc_new = 0;
                unsigned p0 = pl.idx;
                double tfirst = 0.0;
                MP_TICK(tu1);
                MP_ACC(ml, 8, tu0, tu1);
                rmp_genealogy_rest<NM, true, TREES>(cx, t, ml, pl, rp, sb, h, &tc, &tfirst, desc, &desc_new);
                if (cx.vbc) { w_post *= cx.upd_fac; w_pilot *= cx.upd_fac; cx.upd_fac = 1.0; }
                rec[2] = h;
                rec[3] = piece_ref(p0, pl.idx - p0);
                rec[4] = __longlong_as_double((long long)make_meta(0, mark_limit, limit, n, desc, desc_new));
                ++widx;
                if (ml.err) break;
                MP_TICK(tu2);
                if (leaf_status == 0) B = r_tracked_len(t, n, present_mask);
                if (leaf_status == 1) B = cx.Ltree;
                if (biased) {
                    // particle.cpp:866-891: immediate vs delayed application of the importance weight
                    const int nbands = A.n_bias + 1;
                    const double delay_height = (A.delay_type & 3) == 0 ? h : ((A.delay_type & 3) == 2 ? tfirst : tc);
                    int idx = 0;
                    while (idx + 1 < nbands + 1 && sBH[idx + 1] < delay_height) ++idx;
                    if (idx >= nbands) idx = nbands - 1;
                    if (sBS[idx] == 1.0 && !(A.delay_type & 4)) { w_post *= rbiw; w_pilot *= rbiw; iw /= rbiw; }   // bit 2: every factor delayed (pf_model.delay_type)
                    const double delay = A.app_delays[r_epoch_of(cx, delay_height)];
                    d_adjust_with_delay(ds, w_post, w_pilot, iw, delay, updated_to);
                }
                next_base = r_sample_next_base<false>(cx, updated_to);
                x_mark = updated_to;
                mark_limit = limit;
                MP_TICK(tu3);
                MP_ACC(ml, 9, tu2, tu3);
            }
        }
        MP_TICK(tk_loop);
        if (ml.err && !A.ctrl->err) A.ctrl->err = ml.err == 1 ? ERR_MIG_OVERFLOW : (ml.err == 2 ? ERR_MP_INTERNAL : ERR_NO_COALESCENCE);
        if (biased) {
            // apply the factors that fell due during this extension (particle.cpp:910-916)
            for (;;) {
                if (ds.count == 0 || !do_extend) break;
                double pm = ds.pos[0];
                for (int i = 1; i < ds.count; ++i) { double pi = ds.pos[(size_t)i * ds.Np]; if (pi < pm) pm = pi; }
                if (!(pm < extend_to)) break;
                d_apply_earliest(ds, w_pilot);
            }
            st.dcount[p] = ds.count;
            st.total_delayed[p] = ds.total;
            if (guided) st.ridx[p] = cx.ridx;
            has_pending = ds.count > 0;
        }

        if (do_extend && A.seg_state[s] == 0) {
            // update_weight_at_site: marginalise over phasings of unphased hets (pc.cpp:138-224)
            const bool dephase = A.flags & 2;
            const bool anc = A.flags & 1;
            unsigned het_pairs = 0;
            int ncfg = 1;
            for (int i = 0; i + 1 < n; i += 2) {
                const bool d0_two = (two_mask >> i) & 1u;
                const bool one0 = (one_mask >> i) & 1u, one1 = (one_mask >> (i + 1)) & 1u;
                const bool zero0 = (zero_mask >> i) & 1u, zero1 = (zero_mask >> (i + 1)) & 1u;
                const bool het = d0_two || (dephase && ((one0 && zero1) || (zero0 && one1)));
                if (het) {
                    ncfg *= 2;
                    het_pairs |= 1u << i;
                    one_mask &= ~(3u << i); zero_mask &= ~(3u << i);
                    zero_mask |= 1u << i;
                    one_mask |= 1u << (i + 1);
                }
            }
            double norm = 1.0 / (double)ncfg;
            RBranchP<NM> bprob;
            r_site_branch_probs(t, n, A.mu, bprob);     // once per row: the phasing configurations share them
            double lik = 0;
            for (;;) {
                lik += r_site_lik_from(t, n, bprob, one_mask, zero_mask, anc);
                if (ncfg == 1) break;
                bool more = false;
                for (int i = 0; i + 1 < n; i += 2) {
                    if (!((het_pairs >> i) & 1)) continue;
                    if ((zero_mask >> i) & 1) {
                        zero_mask &= ~(1u << i); one_mask |= 1u << i;
                        one_mask &= ~(1u << (i + 1)); zero_mask |= 1u << (i + 1);
                        more = true;
                        break;
                    }
                    one_mask &= ~(1u << i); zero_mask |= 1u << i;
                    zero_mask &= ~(1u << (i + 1)); one_mask |= 1u << (i + 1);
                }
                if (!more) break;
            }
            lik *= norm;
            w_post *= lik;
            w_pilot *= lik;
        }
        MP_TICK(tk_lik);
        MP_ACC(ml, 10, tk_loop, tk_lik);

#pragma unroll
        for (int r = 0; r < NI; ++r)
            if (r < n - 1) {
                st.S[(size_t)r * A.Np + p] = t.S[r];
                st.C[(size_t)(2 * r) * A.Np + p] = (int8_t)t.C0[r];
                st.C[(size_t)(2 * r + 1) * A.Np + p] = (int8_t)t.C1[r];
                st.Pn[(size_t)r * A.Np + p] = (int8_t)pk2_get(ml.pn, r);
                if constexpr (!PIPE) A.snap_S[A.sp][(size_t)r * A.Np + p] = t.S[r];
            }
        st.nm[p] = ml.nm;
        for (int q = 0; q < ml.nm; ++q) {
            st.Mt[(size_t)q * A.Np + p] = LMt(ml, q);
            st.Mb[(size_t)q * A.Np + p] = LMb(ml, q);
            st.Mq[(size_t)q * A.Np + p] = LMq(ml, q);
        }
        st.w_post[p] = w_post;
        st.w_pilot[p] = w_pilot;
        st.next_base[p] = next_base;
        st.x_mark[p] = x_mark;
        st.mark_limit[p] = mark_limit;
        st.Ltree[p] = cx.Ltree;
        A.rng_ctr[p] = cx.ctr;
        A.ebuf[p] = cx.ebuf;
        if constexpr (PIPE) {
            A.rg_widx[(size_t)cur * A.Np + p] = widx;
            if (!do_extend) A.widx[p] = widx;              // the state goes back to the general kernels
            {
                // this launch runs up to PF_RING - 2 rows ahead of the counts: the writer itself checks that it has not overwritten a
                // record that a pending count can still ask for (Ctrl::g_safe: the oldest generation those counts reach)
                const unsigned kold = A.gstart[(size_t)(A.ctrl->g_safe % A.Gcap) * A.Np + p];
                if (widx - kold > A.cap && !A.ctrl->err) A.ctrl->err = ERR_LOG_OVERFLOW;
            }
        } else {
            A.widx[p] = widx;
        }
        A.pidx[p] = pl.idx;
        if (TREES && (widx >= A.cap || pl.idx >= A.pcap)) A.ctrl->err = ERR_LOG_OVERFLOW;       // -arg keeps every record and piece
        if constexpr (!PIPE) { A.snap_w[A.sp][p] = w_post; A.snap_xm[A.sp][p] = x_mark; A.snap_ml[A.sp][p] = mark_limit; A.snap_widx[A.sp][p] = widx; }
        MP_TICK(tk_stored);
        MP_ACC(ml, 11, tk_lik, tk_stored);
        MP_ACC(ml, 15, tk_begin, tk_stored);
    }
#ifdef PF_STAMPS
    __syncthreads();
    if (stamp_out && threadIdx.x < PF_STAMP_W) stamp_out[threadIdx.x] = g_mp_acc[threadIdx.x];
#endif
    if (lane < LA) { sWpost[cslot] = w_post; sWpilot[cslot] = w_pilot; sPend[cslot] = has_pending ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x < 64) {
        // the canonical radix-64 reductions over the workgroup's particles, one per lane
        const long long q = (long long)blockIdx.x * 64 + lane;
        const bool live = q < A.Np;
        const double wq_post = sWpost[lane], wq_pilot = sWpilot[lane];
        double sp = wave_tree_sum(wq_post);
        double sq = wave_tree_sum(wq_pilot * wq_pilot);
        double sc = wave_hs_scan(wq_pilot, lane);
        double scp = wave_hs_scan(wq_post, lane);
        double scm = wave_max_scan_d(sc, lane);
        const long long chunk = blockIdx.x;
        if constexpr (PIPE) {
            const size_t ro = (size_t)cur * A.Np, co = (size_t)cur * A.nc;
            if (live) { A.rg_scan1[ro + q] = sc; A.rg_scanp[ro + q] = scp; A.rg_scan1m[ro + q] = scm; }
            if (q == A.Np - 1) A.ctrl->last1[cur] = sc;
            if (lane == 63) {
                A.rg_cpost[co + chunk] = sp; A.rg_csq[co + chunk] = sq; A.rg_cpil[co + chunk] = sc;
                A.rg_cpp[co + chunk] = scp; A.rg_cmx1[co + chunk] = scm;
            }
            unsigned long long pend = __ballot(sPend[lane] != 0);
            if (lane == 0) {
                A.rg_dpend[co + chunk] = __popcll(pend);
                if (!PR.extend) A.chunk_dpend[chunk] = __popcll(pend);
            }
        } else {
        if (live) { A.scan1[q] = sc; A.scanp2[A.sp][q] = scp; A.scan1m[q] = scm; }
        if (lane == 63) {
            A.chunk_post[chunk] = sp;
            A.chunk_sq[chunk] = sq;
            A.chunk_pil[chunk] = sc;
            A.chunk_pp[chunk] = scp;
            A.chunk_mx1[chunk] = scm;
        }
        if (biased) {
            unsigned long long pend = __ballot(sPend[lane] != 0);
            if (lane == 0) A.chunk_dpend[chunk] = __popcll(pend);
        }
        }
    }
}

template <int NM, bool BIASED, int LA, bool TREES>
__global__ __launch_bounds__(64 * (64 / LA)) void k_extend_mpr(KArgs A, long long s, int fuse) {
    extend_mpr_body<NM, BIASED, LA, TREES, false>(A, s, fuse);
}

// the extend role of the row pipeline for structured models: step t of the chunk table (pf_pipe.h; one chunk per launch
// for now), the bookkeeping / ledger / count roles of the step run as their own launch on the counting stream (pf_hip.hip)
template <int NM, bool BIASED, int LA, bool TREES>
__global__ __launch_bounds__(64 * (64 / LA)) void k_sweep_xmp(const SweepChunk* tab_g, long long t) {
    SweepChunkC* tab = (SweepChunkC*)tab_g;
    SweepChunkC& ch = tab[blockIdx.y];
    KArgsC& A = ch.A;
    const long long s = ch.s_begin + t;
    PipeLaunch PL;
    if (!sweep_plan(ch, s, 0, PL)) return;
    if (PL.row.extend || PL.row.complete) extend_mpr_body<NM, BIASED, LA, TREES, true>(A, s, 0, PL.row);
}

__global__ __launch_bounds__(PF_BS) void k_calibrate_mp(KArgs A, unsigned long long seed, long long rep0, long long nrep,
                                                        int* out_epoch, double* out_dist, int* out_err) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    SmemMP mm = carve_mp(smem, A.n, A.E, A.P, A.mcap);
    load_model(A, m);
    load_model_mp(A, mm);
    __syncthreads();
    long long r = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (r >= nrep) return;
    const int n = A.n;
    Lane ln = make_lane(A, m, rep0 + r);
    MLane ml = make_mlane(A, mm);
    ln.seed = seed;
    ln.stream = 2;
    ln.ebuf = -dlog(uni(ln));
    PLog nolog;
    nolog.on = false;
    mp_build_initial_tree<false>(ln, ml, nolog, [&](int, unsigned, unsigned, double, unsigned) {});
    double* orig = m.t0 + threadIdx.x;
    int alive = n - 1;
    for (int j = 0; j < n - 1; ++j) {
        orig[j * PF_BS] = LS(ln, j);
        out_epoch[r * (n - 1) + j] = epoch_of(ln, LS(ln, j));
        out_dist[r * (n - 1) + j] = -1.0;
    }
    unsigned alive_mask = (1u << (n - 1)) - 1u;
    double next = ml.err ? A.L : sample_next_base(ln, 0.0);
    const double stop = A.L * 0.6;
    while (alive > 0 && next < stop && !ml.err) {
        double x = next;
        int rp = 0, sb = 0;
        double h, tc, sp;
        bool changed;
        sample_point(ln, &rp, &sb, &h);
        mp_genealogy_rest<false>(ln, ml, nolog, -1, rp, sb, h, &tc, &sp, &changed);
        if (ml.err) break;
        if (changed) {
            for (int j = 0; j < n - 1; ++j)
                if (((alive_mask >> j) & 1u) && orig[j * PF_BS] == sp) {
                    out_dist[r * (n - 1) + j] = x;
                    alive_mask &= ~(1u << j);
                    --alive;
                    break;
                }
        }
        next = sample_next_base(ln, x);
    }
    if (ml.err) *out_err = ml.err;
}


// k_simulate for structured models: one lane = one chunk; the prior tree with migration,
// then along the sequence the structured SMC' transition of the filter (mp_genealogy_rest) and, between recombinations,
// mutations dropped on the tree in proportion to branch length.  Its own Philox stream (3), as k_simulate.
__global__ __launch_bounds__(PF_BS) void k_simulate_mp(KArgs A, unsigned long long seed, int nchunks, long long max_sites,
                                                       double* pos_out, unsigned* mask_out, long long* n_out, int* out_err) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    SmemMP mm = carve_mp(smem, A.n, A.E, A.P, A.mcap);
    load_model(A, m);
    load_model_mp(A, mm);
    __syncthreads();
    const long long r = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (r >= nchunks) return;
    Lane ln = make_lane(A, m, r);
    MLane ml = make_mlane(A, mm);
    ln.seed = seed;
    ln.stream = 3;
    ln.ebuf = -dlog(uni(ln));
    PLog nolog;
    nolog.on = false;
    mp_build_initial_tree<false>(ln, ml, nolog, [&](int, unsigned, unsigned, double, unsigned) {});
    double* tmp = m.t0 + threadIdx.x;        // per-lane LDS column for the descendant masks
    double* pos = pos_out + (size_t)r * max_sites;
    unsigned* msk = mask_out + (size_t)r * max_sites;
    long long ns = 0;
    double x = 0.0;
    double next_rec = ml.err ? A.L : sample_next_base(ln, 0.0);
    double next_mut = x + (-dlog(uni(ln))) / (A.mu * ln.Ltree);
    bool overflow = false;
    while (x < A.L && !ml.err) {
        if (next_mut < next_rec && next_mut < A.L) {
            int rp = 0, sb = 0;
            double h;
            sample_point(ln, &rp, &sb, &h);
            const unsigned carriers = lane_desc_mask(ln, LC(ln, rp, sb), tmp);
            if (ns < max_sites) { pos[ns] = next_mut; msk[ns] = carriers; }
            else overflow = true;
            ++ns;
            x = next_mut;
            next_mut = x + (-dlog(uni(ln))) / (A.mu * ln.Ltree);
            continue;
        }
        x = next_rec;
        if (!(x < A.L)) break;
        int rp = 0, sb = 0;
        double h, tc, sp;
        bool changed;
        sample_point(ln, &rp, &sb, &h);
        mp_genealogy_rest<false>(ln, ml, nolog, -1, rp, sb, h, &tc, &sp, &changed);
        next_rec = sample_next_base(ln, x);
        next_mut = x + (-dlog(uni(ln))) / (A.mu * ln.Ltree);     // memoryless: redrawn under the new tree length
    }
    if (ml.err) *out_err = ml.err;
    n_out[r] = overflow ? -ns : ns;
}

// calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166) for a structured model: the parent height of every
// leaf in nrep prior trees (migrations ignored: parent_height_ignoring_migrations, smcsmc.cpp:115-125) and the tree length
__global__ __launch_bounds__(PF_BS) void k_tbl_mp(KArgs A, unsigned long long seed, long long nrep, double* out_h /* [n][nrep] */,
                                                  double* out_len /* [nrep] */, int* out_err) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    SmemMP mm = carve_mp(smem, A.n, A.E, A.P, A.mcap);
    load_model(A, m);
    load_model_mp(A, mm);
    __syncthreads();
    long long r = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (r >= nrep) return;
    const int n = A.n;
    Lane ln = make_lane(A, m, r);
    MLane ml = make_mlane(A, mm);
    ln.seed = seed;
    ln.stream = 3;
    ln.ebuf = -dlog(uni(ln));
    PLog nolog;
    nolog.on = false;
    mp_build_initial_tree<false>(ln, ml, nolog, [&](int, unsigned, unsigned, double, unsigned) {});
    out_len[r] = ln.Ltree;
    for (int i = 0; i < n; ++i) {
        int pr = -1;
        for (int k = 0; k < n - 1 && pr < 0; ++k)
            if (LC(ln, k, 0) == i || LC(ln, k, 1) == i) pr = k;
        out_h[(size_t)i * nrep + r] = pr >= 0 ? LS(ln, pr) : 0.0;
    }
    if (ml.err) *out_err = ml.err;
}

// ------------------------------------------------------------------ launchers (called from pf_hip.hip)
size_t pf_mp_smem_bytes(int n, int E, int P, int mcap) { return smem_bytes_mp(n, E, P, mcap); }

int pf_mp_prepare(size_t smem, int mcap) {
    if (smem > 160 * 1024) return -1;
    {
        // the register-tree row kernels: their LDS does not depend on n (tables at their largest size here)
        const size_t big_sz = smem_mpr_bytes(PF_EMAX, PF_PMAX, mcap);
        const int big = (int)(big_sz < 160 * 1024 ? big_sz : 160 * 1024);
        if (hipFuncSetAttribute((const void*)k_extend_mpr<8, false, PF_MPR_LANES_PLAIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_extend_mpr<8, true, PF_MPR_LANES_BIASED, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_extend_mpr<8, false, PF_MPR_LANES_PLAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_extend_mpr<8, true, PF_MPR_LANES_BIASED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) return -1;
    }
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k_extend_mp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_extend_mp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_init_mp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_calibrate_mp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_tbl_mp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)k_simulate_mp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    }
    return 0;
}
static unsigned mp_blocks(long long n) { return (unsigned)((n + PF_BS - 1) / PF_BS); }

// the extend role of the row pipeline (k_sweep_xmp): LDS = the register-tree kernel's plus the decision's carve-out
size_t pf_mp_sweep_smem_bytes(int E, int P, int mcap, int nc) {
    // the decision's tables (pf_pipe.h) share the LDS of the lanes' migration events: whichever is larger
    const size_t events = (size_t)mcap * PF_BS * 8, tables = pipe_lds_doubles(nc) * 8;
    return smem_mpr_bytes(E, P, mcap) + (tables > events ? tables : 0);
}
int pf_mp_sweep_prepare(size_t smem) {
    if (smem > 160 * 1024) return -1;
    if (hipFuncSetAttribute((const void*)k_sweep_xmp<8, false, PF_MPR_LANES_PLAIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_sweep_xmp<8, true, PF_MPR_LANES_BIASED, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    return 0;
}
void pf_mp_launch_sweep_x(const KArgs& A, const SweepChunk* tab, long long t, size_t smem, hipStream_t st, hipEvent_t done) {
    const bool biased = A.n_bias > 0 || A.g_K > 0;
    const dim3 grid(mp_blocks(A.Np)), blk(64 * (64 / (biased ? PF_MPR_LANES_BIASED : PF_MPR_LANES_PLAIN)));
    if (biased) hipExtLaunchKernelGGL((k_sweep_xmp<8, true, PF_MPR_LANES_BIASED, false>), grid, blk, smem, st, nullptr, done, 0, tab, t);
    else hipExtLaunchKernelGGL((k_sweep_xmp<8, false, PF_MPR_LANES_PLAIN, false>), grid, blk, smem, st, nullptr, done, 0, tab, t);
}
void pf_mp_launch_init(const KArgs& A, double initial_position, size_t smem, hipStream_t st) {
    hipLaunchKernelGGL(k_init_mp, dim3(mp_blocks(A.Np)), dim3(PF_BS), smem, st, A, initial_position);
}
bool pf_mp_can_fuse(const KArgs& A, bool lds_tree) { return !lds_tree && A.n <= 8; }
size_t pf_mp_reg_smem_bytes(int E, int P, int mcap) { return smem_mpr_bytes(E, P, mcap); }
void pf_mp_launch_extend(const KArgs& A, long long s, size_t smem, hipStream_t st, bool lds_tree, int fuse) {
    if (!lds_tree && A.n <= 8) {
        const size_t sm = smem_mpr_bytes(A.E, A.P, A.mcap);
        const bool biased = A.n_bias > 0 || A.g_K > 0;
        const dim3 grid(mp_blocks(A.Np)), blk(64 * (64 / (biased ? PF_MPR_LANES_BIASED : PF_MPR_LANES_PLAIN)));
        if (biased && A.rec_trees) hipLaunchKernelGGL((k_extend_mpr<8, true, PF_MPR_LANES_BIASED, true>), grid, blk, sm, st, A, s, fuse);
        else if (biased) hipLaunchKernelGGL((k_extend_mpr<8, true, PF_MPR_LANES_BIASED, false>), grid, blk, sm, st, A, s, fuse);
        else if (A.rec_trees) hipLaunchKernelGGL((k_extend_mpr<8, false, PF_MPR_LANES_PLAIN, true>), grid, blk, sm, st, A, s, fuse);
        else hipLaunchKernelGGL((k_extend_mpr<8, false, PF_MPR_LANES_PLAIN, false>), grid, blk, sm, st, A, s, fuse);
        return;
    }
    if (A.n_bias > 0 || A.g_K > 0)
        hipLaunchKernelGGL(k_extend_mp<true>, dim3(mp_blocks(A.Np)), dim3(PF_BS), smem, st, A, s);
    else
        hipLaunchKernelGGL(k_extend_mp<false>, dim3(mp_blocks(A.Np)), dim3(PF_BS), smem, st, A, s);
}
void pf_mp_launch_calibrate(const KArgs& A, unsigned long long seed, long long rep0, long long nrep, int* out_epoch,
                            double* out_dist, int* out_err, size_t smem, hipStream_t st) {
    hipLaunchKernelGGL(k_calibrate_mp, dim3(mp_blocks(nrep)), dim3(PF_BS), smem, st, A, seed, rep0, nrep, out_epoch, out_dist,
                       out_err);
}

void pf_mp_launch_tbl(const KArgs& A, unsigned long long seed, long long nrep, double* out_h, double* out_len, int* out_err,
                      size_t smem, hipStream_t st) {
    hipLaunchKernelGGL(k_tbl_mp, dim3(mp_blocks(nrep)), dim3(PF_BS), smem, st, A, seed, nrep, out_h, out_len, out_err);
}

void pf_mp_launch_simulate(const KArgs& A, unsigned long long seed, int nchunks, long long max_sites, double* pos, unsigned* masks,
                           long long* n_sites, int* out_err, size_t smem, hipStream_t st) {
    hipLaunchKernelGGL(k_simulate_mp, dim3(mp_blocks(nchunks)), dim3(PF_BS), smem, st, A, seed, nchunks, max_sites, pos, masks, n_sites, out_err);
}
