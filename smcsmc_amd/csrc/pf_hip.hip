// smcsmc_amd/csrc/pf_hip.hip -- gfx950 kernels + C-ABI (include/smcsmc_pf.h) of the smcsmc particle filter.
//
// One particle per lane.  Per genome segment the stream carries:
//   k_extend   [grid]   ForestState::extend_ARG + site likelihood, per-wavefront partial sums/scans
//   k_decide   [1 WG]   normalisation constant, ESS, resampling decision, offspring table,
//                       backward push of posterior weight through the ancestor ledger
//   k_count    [grid]   CountModel::extract_and_update_count over the per-slot event logs
//   k_count_fin[1 WG]   ordered reduction of k_count partials
//   k_resample [grid]   normalisation or systematic-resampling gather (ParticleContainer::resample)
// The host never reads device memory inside pf_run: the decision to resample is taken on the
// device and every kernel is launched unconditionally (k_count only on segments where the
// reference's lag rule, which is particle-independent, schedules an update).
//
// No CUDA compatibility layer, no CPU fallback: this file is HIP for gfx950 only.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smcsmc_pf.h"
#include "pf_device.h"
#include "pf_types.h"
#include "pf_lane.h"
#include "pf_tree_reg.h"
#include "pf_mp_host.h"
#include "pf_pipe.h"

using namespace pf;

// ------------------------------------------------------------------ k_init  (particleContainer.cpp:33-65)
__global__ __launch_bounds__(PF_BS) void k_init(KArgs A, double initial_position) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    load_model(A, m);
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
    ln.ebuf = -dlog(uni(ln));
    unsigned widx = 0;
    int root = 0;
    double w0 = 1.0 / (double)A.Np;
    // Forest::buildInitialTree(true): add the samples one by one; every coalescence is logged
    // as a type-2 record at position 0 (record_all_event, particle.cpp:251-300)
    for (int i = 1; i < n; ++i) {
        int ni = i - 1;
        double* rec = rec_ptr(A, p, widx);
        rec[0] = 0.0; rec[1] = 0.0; rec[2] = 0.0;
        for (int r = 0; r < n - 1; ++r) rec[5 + r] = r < ni ? LS(ln, r) : 0.0;
        double tc = coalesce_up(ln, [&](int k) { return LS(ln, k); }, ni, i, 0.0);
        if (ln.vbc) { w0 *= ln.upd_fac; ln.upd_fac = 1.0; }
        rec[3] = tc;
        ++widx;
        int pr = -1, ps = 0;
        int k = lineages_at(ln, ni, tc, -1, &pr, &ps);
        bool above_root = (ni == 0) || (tc >= LS(ln, ni - 1));
        int kk = above_root ? 1 : k;
        double u = uni(ln);
        int idx = min((int)(u * (double)kk), kk - 1);
        unsigned dn = (2u << i) - 1u;                 // above the root: all samples added so far
        if (above_root) {
            insert_node(ln, ni, tc, i, -1, 0, root);
        } else {
            lineages_at(ln, ni, tc, idx, &pr, &ps);
            if (A.rec_trees) dn = (1u << i) | lane_desc_mask(ln, LC(ln, pr, ps), m.t0 + threadIdx.x);
            insert_node(ln, ni, tc, i, pr, ps, root);
        }
        rec[4] = __longlong_as_double((long long)make_meta(2, A.E - 1, A.E - 1, i, 0, A.rec_trees ? dn : 0u));
        root = n + ni;
    }
    ln.Ltree = tree_length(ln, n);
    if (A.g_K > 0) {
        // with a guide the first draw uses the rate of the first segment and stops at its end
        ln.rho = A.g_rho[0];
        if (A.g_K > 1 && A.g_pos[1] < A.L) ln.L = A.g_pos[1];
    }
    double nb = sample_next_base(ln, 0.0);
    const DState st = A.st0;
    for (int r = 0; r < n - 1; ++r) {
        st.S[(size_t)r * A.Np + p] = LS(ln, r);
        st.C[(size_t)(2 * r) * A.Np + p] = LC(ln, r, 0);
        st.C[(size_t)(2 * r + 1) * A.Np + p] = LC(ln, r, 1);
    }
    st.w_post[p] = w0;
    st.w_pilot[p] = w0;
    st.next_base[p] = nb;
    st.x_mark[p] = 0.0;
    st.Ltree[p] = ln.Ltree;
    st.mark_limit[p] = A.E - 1;
    if (A.n_bias > 0 || A.g_K > 0) { st.total_delayed[p] = 1.0; st.dcount[p] = 0; }
    if (A.g_K > 0) st.ridx[p] = 0;
    if (st.lookahead) st.lookahead[p] = 1.0;
    A.rng_ctr[p] = ln.ctr;
    A.ebuf[p] = ln.ebuf;
    A.widx[p] = widx;
    A.gstart[p] = 0;   // generation 0 starts with an empty log (the init records belong to it)
}

// ------------------------------------------------------------------ k_extend
// ParticleContainer::extend_ARGs + update_weight_at_site (particleContainer.cpp:98-135, 187-224)
// with ForestState::extend_ARG (particle.cpp:743-918) per lane.
__global__ __launch_bounds__(PF_BS) void k_extend(KArgs A, long long s) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    load_model(A, m);
    __shared__ double sBH[PF_BIAS_MAX + 2], sBS[PF_BIAS_MAX + 1];      // focused sampling: band boundaries / strengths
    if (threadIdx.x < PF_BIAS_MAX + 2) {
        sBH[threadIdx.x] = A.bias_H[threadIdx.x];
        if (threadIdx.x < PF_BIAS_MAX + 1) sBS[threadIdx.x] = A.bias_S[threadIdx.x];
    }
    __syncthreads();
    const bool guided = A.g_K > 0;
    const bool biased = A.n_bias > 0 || guided;          // a guide alone runs with one band of strength 1
    bool has_pending = false;
    const Ctrl* c = A.ctrl;
    const int n = A.n;
    const int cur = __builtin_amdgcn_readfirstlane(c->cur);
    const long long p = (long long)blockIdx.x * PF_BS + threadIdx.x;
    const bool active = p < A.Np;
    const int lane = threadIdx.x & 63;
    double w_post = 0.0, w_pilot = 0.0;
    if (active) {
        const DState st = state_slot(A, cur);
        Lane ln = make_lane(A, m, p);
        for (int r = 0; r < n - 1; ++r) {
            LS(ln, r) = st.S[(size_t)r * A.Np + p];
            LC(ln, r, 0) = st.C[(size_t)(2 * r) * A.Np + p];
            LC(ln, r, 1) = st.C[(size_t)(2 * r + 1) * A.Np + p];
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
        DStore ds;
        d_bind(ds, A, st, p);
        ds.count = 0; ds.total = 1.0;
        if (biased) { ds.count = st.dcount[p]; ds.total = st.total_delayed[p]; }
        int ridx = guided ? st.ridx[p] : 0;
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

        while (updated_to < extend_to) {
            double new_to = extend_to < next_base ? extend_to : next_base;
            double f = fastexp(-A.mu * B * (new_to - updated_to));
            w_post *= f;
            w_pilot *= f;
            if (guided) {
                // importance_weight_over_segment (particle.cpp:1138-1181): true over guide rate for the stretch
                // without recombination
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
                // a recombination: log the stretch that ends here together with the event
                double* rec = rec_ptr(A, p, widx);
                rec[0] = x_mark;
                rec[1] = updated_to;
                for (int r = 0; r < n - 1; ++r) rec[5 + r] = LS(ln, r);
                double h, tc, sp_removed;
                bool changed;
                unsigned desc = 0, desc_new = 0;
                double iw = 1.0, rbiw = 1.0;
                genealogy_update(ln, &h, &tc, &sp_removed, &changed, (A.lmap_opp || A.rec_trees) ? &desc : nullptr, tmp0,
                                 biased ? sBH : nullptr, sBS, A.n_bias + 1, &iw,
                                 guided ? A.g_leaf + (size_t)ridx * n : nullptr, guided ? A.rho / A.g_rho[ridx] : 1.0, &rbiw,
                                 A.rec_trees ? &desc_new : nullptr);
                if (ln.vbc) { w_post *= ln.upd_fac; w_pilot *= ln.upd_fac; ln.upd_fac = 1.0; }
                rec[2] = h;
                rec[3] = tc;
                rec[4] = __longlong_as_double((long long)make_meta(0, mark_limit, limit, n, desc, desc_new));
                ++widx;
                if (leaf_status == 0) B = tracked_len_lane(ln, data, tmp0);
                if (leaf_status == 1) B = ln.Ltree;
                if (biased) {
                    // particle.cpp:866-891: immediate vs delayed application of the importance weight
                    const int nbands = A.n_bias + 1;
                    const double delay_height = (A.delay_type & 3) == 0 ? h : tc;
                    int idx = 0;
                    while (idx + 1 < nbands + 1 && sBH[idx + 1] < delay_height) ++idx;
                    if (idx >= nbands) idx = nbands - 1;
                    if (sBS[idx] == 1.0 && !(A.delay_type & 4)) { w_post *= rbiw; w_pilot *= rbiw; iw /= rbiw; }   // bit 2: every factor delayed (pf_model.delay_type)
                    const double delay = A.app_delays[epoch_of(ln, delay_height)];
                    d_adjust_with_delay(ds, w_post, w_pilot, iw, delay, updated_to);
                }
                next_base = sample_next_base_guided(ln, updated_to, A.g_K, A.g_pos, A.g_rho, ridx);
                ln.uqn = 0;                    // the update's unused uniforms are dropped
                x_mark = updated_to;
                mark_limit = limit;
            }
        }

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
                    zero_mask |= 1u << i;          // hap[i] = 0
                    one_mask |= 1u << (i + 1);     // hap[i+1] = 1
                }
            }
            double norm = 1.0 / (double)ncfg;
            double lik = 0;
            for (;;) {
                lik += site_lik_lane(ln, one_mask, zero_mask, anc, tmp0, tmp1);
                if (ncfg == 1) break;
                bool more = false;                  // next_haplotype (pc.cpp:163-181)
                for (int i = 0; i + 1 < n; i += 2) {
                    if (!((het_pairs >> i) & 1)) continue;
                    if ((zero_mask >> i) & 1) {     // phase 0 -> phase 1
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

        for (int r = 0; r < n - 1; ++r) {
            st.S[(size_t)r * A.Np + p] = LS(ln, r);
            st.C[(size_t)(2 * r) * A.Np + p] = LC(ln, r, 0);
            st.C[(size_t)(2 * r + 1) * A.Np + p] = LC(ln, r, 1);
        }
        st.w_post[p] = w_post;
        st.w_pilot[p] = w_pilot;
        st.next_base[p] = next_base;
        st.x_mark[p] = x_mark;
        st.mark_limit[p] = mark_limit;
        st.Ltree[p] = ln.Ltree;
        A.rng_ctr[p] = ln.ctr;
        A.ebuf[p] = ln.ebuf;
        A.widx[p] = widx;
        if (A.rec_trees && widx >= A.cap) A.ctrl->err = ERR_LOG_OVERFLOW;
        for (int r = 0; r < n - 1; ++r) A.snap_S[A.sp][(size_t)r * A.Np + p] = LS(ln, r);
        A.snap_w[A.sp][p] = w_post; A.snap_xm[A.sp][p] = x_mark; A.snap_ml[A.sp][p] = mark_limit; A.snap_widx[A.sp][p] = widx;
    }
    // per-wavefront canonical partials (level 1 of the radix-64 reduction / scan)
    double sp = wave_tree_sum(w_post);
    double sq = wave_tree_sum(w_pilot * w_pilot);
    double sc = wave_hs_scan(w_pilot, lane);
    double scp = wave_hs_scan(w_post, lane);
    double scm = wave_max_scan_d(sc, lane);     // running max of the pilot scan (a parallel FP scan need not be monotone)
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

// A wave-uniform value moved into vector registers on purpose.  The extend kernels are short of scalar registers (the
// argument block alone is ~240 of them; spilled ones come back through v_readlane inside the update loop, with the
// s_nop padding that follows), while only a third of the vector file is in use: the handful of scalars the update loop
// reads on every trip live in VGPRs instead.
__device__ __forceinline__ double in_vgpr(double x) { double y; asm("v_mov_b64 %0, %1" : "=v"(y) : "s"(x)); return y; }
__device__ __forceinline__ unsigned long long in_vgpr(unsigned long long x) { unsigned long long y; asm("v_mov_b64 %0, %1" : "=v"(y) : "s"(x)); return y; }

#ifdef PF_STAMPS
__device__ __forceinline__ void pf_acc_trip(unsigned long long* a) { a[5] += 1; }
#else
#define pf_acc_trip(a) do {} while (0)
#endif
#ifdef PF_STAMPS
#define PF_STAMP(k) do { if (A.stamps && (threadIdx.x & 63) == 0 && s < A.stamp_rows)                                   \
        A.stamps[((size_t)s * A.nc + (size_t)(((long long)pf_bx() * PF_BS + threadIdx.x) >> 6)) * PF_STAMP_W + (k)] = wall_clock64(); } while (0)
#define PF_TICK(var) unsigned long long var = wall_clock64()
#define PF_ACC(slot, t0, t1) pf_acc[slot] += (t1) - (t0)
#define PF_ACC_DECL unsigned long long pf_acc[6] = {0, 0, 0, 0, 0, 0}
#define PF_ACC_STORE do { if (A.stamps && (threadIdx.x & 63) == 0 && s < A.stamp_rows)                                  \
        for (int k_ = 0; k_ < 6; ++k_)                                                                                  \
            A.stamps[((size_t)s * A.nc + (size_t)(((long long)pf_bx() * PF_BS + threadIdx.x) >> 6)) * PF_STAMP_W + 9 + k_] = pf_acc[k_]; } while (0)
#else
#define PF_STAMP(k) do {} while (0)
#define PF_TICK(var) do {} while (0)
#define PF_ACC(slot, t0, t1) do {} while (0)
#define PF_ACC_DECL do {} while (0)
#define PF_ACC_STORE do {} while (0)
#endif

template <class KA>
__device__ __forceinline__ int gridDim_particles(const KA& A) { return (int)((A.Np + PF_BS - 1) / PF_BS); }

// ------------------------------------------------------------------ k_extend_reg
// Same computation as k_extend with the local tree held in registers (pf_tree_reg.h); used for
// n <= 8.  LDS only carries the two epoch tables.
// With fuse != 0 the launch also completes the previous row: the in-place normalisation or the resampling gather of
// k_resample (pc.cpp:321-392, 435-437) happens while the particle is loaded, which removes one kernel and its
// launch gap from the per-row critical path.  The arithmetic is k_resample's, operation for operation.
// EXACT: the number of haplotypes equals NM, so every `r < n - 1` guard of the unrolled tree loops is decided at
// compile time (the guards are compare + exec-mask instructions, and instructions are what the time is made of)
// TREES: -arg, the records also carry the descendants of the node each update creates
// PIPE: single-launch pipeline (k_pipe).  The workgroup takes the decision on the previous row itself (decide_row),
// derives the offspring offsets of its own particles and finds the parent of each of its slots by a two-level search
// over the pilot prefix sums -- no offspring / parent table from an earlier kernel is read --, reads the previous row
// from one slot of the state ring and writes this row into the next, and leaves the offspring table for the ledger.
template <int NM, bool BIASED, bool EXACT = false, bool TREES = false, bool PIPE = false, class KA>
__device__ __forceinline__ void extend_reg_body(const KA& A, long long s, int fuse, PipeRow PR = PipeRow()) {
    extern __shared__ double smem[];
    double* sT = smem;                            // epoch starts and ...
    double* sH = smem + PF_EPAD;                  // ... cumulative coalescence intensity there, both padded with +inf (r_search4)
    double* sI = sH + PF_EPAD;
    double* sBH = sI + A.E;                       // bias band boundaries / strengths (focused sampling)
    double* sBS = sBH + (PF_BIAS_MAX + 2);
    const Ctrl* c = A.ctrl;
    const int n = EXACT ? NM : A.n;
    const long long p = (long long)pf_bx() * PF_BS + threadIdx.x;
    const bool active = p < A.Np;
    const int lane = threadIdx.x & 63;
    // PIPE: everything the prologue and the particle itself need from memory is requested first, in one round trip: the
    // partials of the decision, this slot's own state of the previous row (what most rows continue from, and what the
    // old slot's closing record needs when the row resampled) -- a copy reloads from its parent afterwards.
    PF_STAMP(0);
    RowPre pre;
    RTree<NM> t0;
    double o_wpost = 0.0, o_wpilot = 0.0, o_next = 0.0, o_xmark = 0.0, o_Ltree = 0.0, o_sm = 0.0, o_ebuf = 0.0;
    int o_ml = 0;
    unsigned o_widx = 0;
    unsigned long long o_ctr = 0;
    unsigned o_tab_end = 0;
    long long o_nres = 0; int o_bflag = 0, o_bgen = 0;
    double spec_sm[PF_PIPE_STAGE * 64 / PF_BS];
    int spec_lo = 0;
    if constexpr (PIPE) {
        const int fs = __builtin_amdgcn_readfirstlane(PR.slot_prev >= 0 ? PR.slot_prev : c->cur);
        if (PR.complete) {
            pre = row_preload(A, fs);
            // the row before it: what the extend role of the previous launch noted about it (the same numbers the bookkeeping
            // role publishes in Ctrl::ri, without depending on the launch that carries that role)
            o_nres = c->xr[(fs + PF_RING - 1) & (PF_RING - 1)].n_res; o_bflag = c->xr[(fs + PF_RING - 1) & (PF_RING - 1)].flag; o_bgen = c->xr[(fs + PF_RING - 1) & (PF_RING - 1)].gen;
            // The parent search of a resampling row stages the pilot scans of the wavefronts its parents sit in.  Offspring
            // stay close to their parents' slots (the offsets drift like a random walk of a few hundred slots), so the scans
            // of this workgroup's own wavefronts and six on either side are requested now, with everything else the prologue
            // reads, instead of in a round trip of their own once the search knows the range; a range outside this window
            // is staged the old way.
            spec_lo = pf_bx() * (PF_BS / 64) - (PF_PIPE_STAGE - PF_BS / 64) / 2;
            if (spec_lo > A.nc - PF_PIPE_STAGE) spec_lo = A.nc - PF_PIPE_STAGE;
            if (spec_lo < 0) spec_lo = 0;
            if (A.flags & 256) spec_lo = -0x40000000;              // PF_DEBUG_NO_SPEC_STAGE (A/B): never covers the range
            const double* sm0 = A.rg_scan1m + (size_t)fs * A.Np;
#pragma unroll
            for (int k = 0; k < PF_PIPE_STAGE * 64 / PF_BS; ++k) {
                const long long src = (long long)spec_lo * 64 + k * PF_BS + threadIdx.x;
                spec_sm[k] = (src >= 0 && src < A.Np) ? sm0[src] : PF_INF;
            }
        }
        if (active) {
            const DState own = state_slot(A, fs);
#pragma unroll
            for (int r = 0; r < RTree<NM>::NI; ++r) {
                t0.S[r] = 0.0; t0.C0[r] = 0; t0.C1[r] = 0;
                if (r < n - 1) {
                    t0.S[r] = own.S[(size_t)r * A.Np + p];
                    t0.C0[r] = own.C[(size_t)(2 * r) * A.Np + p];
                    t0.C1[r] = own.C[(size_t)(2 * r + 1) * A.Np + p];
                }
            }
            o_wpost = own.w_post[p]; o_wpilot = own.w_pilot[p]; o_next = own.next_base[p]; o_xmark = own.x_mark[p];
            o_ml = own.mark_limit[p]; o_Ltree = own.Ltree[p];
            o_widx = PR.complete ? A.rg_widx[(size_t)fs * A.Np + p] : A.widx[p];
            o_ctr = A.rng_ctr[p]; o_ebuf = A.ebuf[p];
            if (PR.draws) o_tab_end = (unsigned)A.dt_filled[(size_t)((PR.draws - 1) ^ 1) * A.Np + p];
            if (PR.complete) o_sm = A.rg_scan1m[(size_t)fs * A.Np + p];
        }
    }
    for (int e = threadIdx.x; e < PF_EPAD; e += blockDim.x) {
        sT[e] = e < A.E ? A.T[e] : PF_INF;
        sH[e] = e < A.E ? A.Hc[e] : PF_INF;
        if (e < A.E) sI[e] = A.inv2N[e];
    }
    if (BIASED && threadIdx.x < PF_BIAS_MAX + 2) {
        sBH[threadIdx.x] = A.bias_H[threadIdx.x];
        if (threadIdx.x < PF_BIAS_MAX + 1) sBS[threadIdx.x] = A.bias_S[threadIdx.x];
    }
    __shared__ unsigned s_lut[PIPE ? 2 * PF_LUT_N / 4 : 1];
    const bool use_lut = PIPE && A.lut != nullptr;
    if constexpr (PIPE) {
        if (use_lut && threadIdx.x < 2 * PF_LUT_N / 4) s_lut[threadIdx.x] = ((const unsigned*)A.lut)[threadIdx.x];
    }
    // ---- what completing the previous row needs ----
    int cur = 0, from_slot = 0;
    bool completing = false, gather = false;
    double inv = 1.0, S1v = 0.0, pos_prev = 0.0;
    int lo_p = 0, lo_p1 = 0;
    bool first_copy = true;
    long long a = p;
    int G_end = 0, ev = 0;
    if constexpr (PIPE) {
        PipeLds q = pipe_carve(sBS + (PF_BIAS_MAX + 1), A.nc);
        cur = PR.slot_out; from_slot = PR.slot_prev >= 0 ? PR.slot_prev : c->cur;
        completing = PR.complete != 0;
        pos_prev = PR.pos_prev;
        if (completing) {
            const int row_slot = from_slot;                        // ring slot of the row being completed
            const long long n_res = o_nres + o_bflag;
            G_end = o_bgen + o_bflag;
            ev = (int)n_res;
            PF_STAMP(1);
#pragma unroll
            for (int k = 0; k < PF_PIPE_STAGE * 64 / PF_BS; ++k) q.stage[k * PF_BS + threadIdx.x] = spec_sm[k];    // visible after decide_row's barriers
            RowDecision d = decide_row<true>(A, q, row_slot, n_res, pre);
            PF_STAMP(2);
            inv = d.inv; S1v = d.S1;
            gather = d.flag != 0;
            if (pf_bx() == 0 && threadIdx.x == 0) {
                Ctrl* cw = A.ctrl;
                cw->xr[row_slot].n_res = n_res; cw->xr[row_slot].gen = G_end; cw->xr[row_slot].flag = d.flag;
            }
            if (gather) {
                const double dn = (double)A.Np;
                const double invS1 = 1.0 / d.S1;
                const double* sm = A.rg_scan1m + (size_t)row_slot * A.Np;
                const int wave = threadIdx.x >> 6;
                int ch_own = (int)(p >> 6);
                double coff_own = 0.0, v_own = 0.0;
                int lo_next = 0;
                if (active) {
                    coff_own = pipe_chunk_offset(q, ch_own);
                    double w = coff_own + o_sm;
                    v_own = q.pmx[ch_own] > w ? q.pmx[ch_own] : w;          // largest pilot prefix sum up to particle p
                    lo_next = p + 1 < A.Np ? pipe_lo_from(v_own, dn, A.Np, d.S1, invS1, d.u) : (int)A.Np;
                    q.slo[threadIdx.x] = lo_next;
                }
                PF_STAMP(16);
                // ---- parent of slot p: the first particle a with (p+u) * S1 < N * (largest prefix sum up to a) ----
                const double lhs = ((double)p + d.u) * d.S1;
                int pch = 0;
                if (active) {
                    int lo_c = 0, hi_c = A.nc - 1;
                    while (lo_c < hi_c) {
                        int mid = (lo_c + hi_c) >> 1;
                        if (lhs < dn * q.pmx[mid + 1]) hi_c = mid; else lo_c = mid + 1;
                    }
                    pch = lo_c;
                }
                PF_STAMP(17);
                int cmin = active ? pch : 0x7fffffff, cmax = active ? pch : -1;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    int o1 = __shfl_xor(cmin, m, 64), o2 = __shfl_xor(cmax, m, 64);
                    cmin = o1 < cmin ? o1 : cmin; cmax = o2 > cmax ? o2 : cmax;
                }
                if (lane == 0) { q.wint[wave] = cmin; q.wint[PF_BS / 64 + wave] = cmax; }
                __syncthreads();
                if (active) lo_p1 = lo_next;
                if (active) {
                    if (threadIdx.x > 0) lo_p = q.slo[threadIdx.x - 1];
                    else lo_p = p > 0 ? pipe_lo_from(q.pmx[ch_own], dn, A.Np, d.S1, invS1, d.u) : 0;
                }
                cmin = q.wint[0]; cmax = q.wint[PF_BS / 64];
                for (int w = 1; w < PF_BS / 64; ++w) {
                    cmin = q.wint[w] < cmin ? q.wint[w] : cmin;
                    cmax = q.wint[PF_BS / 64 + w] > cmax ? q.wint[PF_BS / 64 + w] : cmax;
                }
                // survivors of this workgroup (the ledger positions the run list of the ending generation with them)
                {
                    unsigned long long bal = __ballot(active && lo_p1 > lo_p);
                    if (lane == 0) q.wint[2 * (PF_BS / 64) + wave] = __popcll(bal);
                }
                PF_STAMP(18);
                int nst = cmax - cmin + 1;
                if (nst > PF_PIPE_STAGE) nst = PF_PIPE_STAGE;
                if (nst < 0) nst = 0;
                int stage_lo = cmin;
                if (cmin >= spec_lo && cmax < spec_lo + PF_PIPE_STAGE) {
                    stage_lo = spec_lo; nst = PF_PIPE_STAGE;       // the scans requested with the prologue cover the range
                } else {
                    __syncthreads();                               // everybody has read the range before the buffer is refilled
                    for (int idx = threadIdx.x; idx < nst * 64; idx += PF_BS) {
                        long long src = (long long)cmin * 64 + idx;
                        q.stage[idx] = src < A.Np ? sm[src] : PF_INF;
                    }
                }
                __syncthreads();
                PF_STAMP(19);
                if (threadIdx.x == 0) {
                    int tot = 0;
                    for (int w = 0; w < PF_BS / 64; ++w) tot += q.wint[2 * (PF_BS / 64) + w];
                    A.rg_blkcnt[(size_t)row_slot * gridDim_particles(A) * A.blk_gran + pf_bx()] = tot;
                }
                if (active) {
                    const double coff_p = pipe_chunk_offset(q, pch);
                    const double pm_p = q.pmx[pch];
                    const bool staged = pch >= stage_lo && pch - stage_lo < nst;
                    auto val_at = [&](int l) -> double {           // largest prefix sum up to particle pch*64 + l
                        long long idx = (long long)pch * 64 + l;
                        double smv = staged ? q.stage[(pch - stage_lo) * 64 + l] : (idx < A.Np ? sm[idx] : PF_INF);
                        double w = coff_p + smv;
                        return pm_p > w ? pm_p : w;
                    };
                    int lo_l = 0, hi_l = 63;
                    while (lo_l < hi_l) {
                        int mid = (lo_l + hi_l) >> 1;
                        if (lhs < dn * val_at(mid)) hi_l = mid; else lo_l = mid + 1;
                    }
                    a = (long long)pch * 64 + lo_l;
                    PF_STAMP(20);
                    if (a > A.Np - 1) a = A.Np - 1;
                    // slot p is the first copy of a (it keeps a's next recombination position) iff it equals a's offset
                    const int la = (int)(a & 63);
                    if (p == 0 || a == 0) first_copy = (p == 0);
                    else {
                        double vprev = la > 0 ? val_at(la - 1) : pm_p;      // largest prefix sum up to a - 1
                        if (a != (long long)pch * 64 + lo_l) {             // clamped: recompute on the true chunk of a - 1
                            long long am = a - 1;
                            int chm = (int)(am >> 6);
                            double w = pipe_chunk_offset(q, chm) + sm[am];
                            vprev = q.pmx[chm] > w ? q.pmx[chm] : w;
                        }
                        first_copy = ((((double)(p - 1)) + d.u) * d.S1 < dn * vprev);
                    }
                    // the offspring table of the generation that ends here, for the ledger (launch s + 1) and -arg
                    int* lo_tab = A.lo + (size_t)(G_end % A.Gcap) * (A.Np + 1);
                    lo_tab[p] = lo_p;
                    if (p == A.Np - 1) lo_tab[A.Np] = (int)A.Np;
                }
            }
        }
    } else {
        cur = c->cur;
        completing = fuse != 0;
        gather = fuse && c->flag;
        from_slot = gather ? (cur ^ 1) : cur;
        __syncthreads();
    }
    if constexpr (PIPE) __syncthreads();
    PF_STAMP(3);
    // wave-uniform slots: indexed kernel arguments stay scalar loads (a lane-varying index would put them on the stack)
    cur = __builtin_amdgcn_readfirstlane(cur);
    from_slot = __builtin_amdgcn_readfirstlane(from_slot);
    double w_post = 0.0, w_pilot = 0.0;
    bool has_pending = false;
    if (active) {
        const DState st = state_slot(A, cur);
        const DState from = state_slot(A, from_slot);
        if constexpr (!PIPE) {
            a = gather ? (long long)A.parent[p] : p;
            // offspring offsets of this slot (as the old particle) and of its parent: requested with the parent index and
            // with the parent's state, not after them (each dependent request is a memory round trip of about 1 us)
            const int* lo_tab = A.lo + (size_t)((c->gen - 1 + A.Gcap) % A.Gcap) * (A.Np + 1);
            if (gather) { lo_p = lo_tab[p]; lo_p1 = lo_tab[p + 1]; first_copy = (p == lo_tab[a]); }
            inv = c->inv_T; S1v = c->S1; pos_prev = c->cur_pos;
            G_end = c->gen - 1; ev = (int)c->n_resample - 1;
        }
        RTree<NM> t;
        const bool reload = !PIPE || gather;           // PIPE: the slot's own state is already in registers
        if (!reload) t = t0;
        else {
#pragma unroll
            for (int r = 0; r < RTree<NM>::NI; ++r) {
                t.S[r] = 0.0; t.C0[r] = 0; t.C1[r] = 0;
                if (r < n - 1) {
                    t.S[r] = from.S[(size_t)r * A.Np + a];
                    t.C0[r] = from.C[(size_t)(2 * r) * A.Np + a];
                    t.C1[r] = from.C[(size_t)(2 * r + 1) * A.Np + a];
                }
            }
        }
        RCtx cx;
        const double v_L = in_vgpr(A.L), v_mu = in_vgpr(A.mu), v_rho = in_vgpr(A.rho);
        cx.T = sT; cx.I = sI; cx.H = sH; cx.E = A.E; cx.n = n; cx.L = v_L; cx.mu = v_mu; cx.rho = v_rho;
        cx.seed = in_vgpr(A.seed); cx.slot = (unsigned)p; cx.stream = 0;
        cx.nb = A.n_bias + 1; cx.bH = sBH; cx.bS = sBS; cx.last_iw = 1.0;
        cx.want_desc = A.lmap_opp != nullptr || TREES; cx.want_desc_new = TREES; cx.last_desc = 0; cx.last_desc_new = 0;
        cx.vbc = A.vb_coal; cx.upd_fac = 1.0;
        cx.gK = BIASED ? A.g_K : 0; cx.gpos = A.g_pos; cx.grho = A.g_rho; cx.gleaf = A.g_leaf; cx.last_rbiw = 1.0;
        cx.ridx = cx.gK > 0 ? from.ridx[a] : 0; cx.g_rp = 0; cx.g_sb = 0;
        if constexpr (PIPE) {
            cx.lutT = use_lut ? (const unsigned char*)s_lut : nullptr;
            cx.lutH = use_lut ? (const unsigned char*)s_lut + PF_LUT_N : nullptr;
            cx.kbT = A.lut_kbT; cx.kbH = A.lut_kbH;
            cx.tab = nullptr; cx.tab_end = 0; cx.pf_ok = false; cx.pf_ctr = 0; cx.draws_log = false;
            if (PR.draws) {
                cx.tab = (const double2*)A.dt_tab + (size_t)p * PF_DRAW_RING;
                cx.tab_end = o_tab_end;
            }
        }
        DStore ds;
        if (BIASED) {
            d_bind(ds, A, st, p);
            ds.count = from.dcount[a]; ds.total = from.total_delayed[a];
            if (gather || PIPE)   // the copy constructor copies the pending factors (particle.cpp:122-123); the ring moves them every row
                for (int k = 0; k < ds.count; ++k) {
                    st.dpos[(size_t)k * A.Np + p] = from.dpos[(size_t)k * A.Np + a];
                    st.dfac[(size_t)k * A.Np + p] = from.dfac[(size_t)k * A.Np + a];
                    st.ddelta[(size_t)k * A.Np + p] = from.ddelta[(size_t)k * A.Np + a];
                    st.dk[(size_t)k * A.Np + p] = from.dk[(size_t)k * A.Np + a];
                }
        }
        double next_base, x_mark;
        int mark_limit;
        unsigned widx;
        if (reload) {
            w_post = from.w_post[a];
            w_pilot = from.w_pilot[a];
            next_base = from.next_base[a];
            x_mark = from.x_mark[a];
            mark_limit = from.mark_limit[a];
            cx.Ltree = from.Ltree[a];
        } else {
            w_post = o_wpost; w_pilot = o_wpilot; next_base = o_next; x_mark = o_xmark; mark_limit = o_ml; cx.Ltree = o_Ltree;
        }
        if constexpr (PIPE) { cx.ctr = o_ctr; cx.ebuf = o_ebuf; widx = o_widx; r_draws_prefetch(cx); }
        else { cx.ctr = A.rng_ctr[p]; cx.ebuf = A.ebuf[p]; widx = A.widx[p]; }
        if (completing) {
            if (!PIPE && p == 0) { Ctrl* cw = A.ctrl; cw->gen_prev = c->gen; cw->nres_prev = c->n_resample; }
            if (!gather) {
                w_post *= inv;                             // normalize_probability, pc.cpp:435-437
                w_pilot *= inv;
            } else {
                const int G = G_end;                       // the generation that ended with the previous row
                const double pos = pos_prev;
                const DState& src = from;
                // role of the old slot p: close its stretch if it has offspring
                if (lo_p1 > lo_p) {
                    double* rec = rec_ptr(A, p, widx);
                    rec[0] = PIPE ? o_xmark : src.x_mark[p];
                    rec[1] = pos;
                    rec[2] = 0.0; rec[3] = 0.0;
                    rec[4] = __longlong_as_double((long long)make_meta(1, PIPE ? o_ml : src.mark_limit[p], -1, n));
                    if constexpr (PIPE) {
#pragma unroll
                        for (int r = 0; r < RTree<NM>::NI; ++r) if (r < n - 1) rec[5 + r] = t0.S[r];
                    } else {
                        for (int r = 0; r < n - 1; ++r) rec[5 + r] = src.S[(size_t)r * A.Np + p];
                    }
                    ++widx;
                }
                A.gstart[(size_t)((G + 1) % A.Gcap) * A.Np + p] = widx;
                // role of the new slot p: weights of the copy (pc.cpp:350-351), fresh position for all but the first
                if (ev < A.max_trace_events) A.ev_parents[(size_t)ev * A.Np + p] = (int)a;
                double wp = w_post * inv;
                double wq = w_pilot * inv;
                double sumn = S1v * inv;
                double adj = sumn / ((double)A.Np * wq);
                w_post = wp * adj;
                w_pilot = wq * adj;
                x_mark = pos;
                if (!first_copy && pos < v_L) {
                    next_base = r_sample_next_base<false>(cx, pos);     // pc.cpp:357-368
                    if constexpr (PIPE) r_draws_prefetch(cx);           // the counter moved: the request of the prologue is stale
                }
            }
        }

        PF_STAMP(4);
        const bool do_extend = !PIPE || PR.extend != 0;
        const int8_t* data = A.seg_alleles + (size_t)s * n;
        const double seg_end = do_extend ? A.seg_start[s] + A.seg_len[s] : 0.0;
        const double extend_to = seg_end < v_L ? seg_end : v_L;
        const int limit = do_extend ? A.seg_limit[s] : 0;
        unsigned one_mask = 0, zero_mask = 0, present_mask = 0, two_mask = 0;
        int missing = 0;
        if (do_extend)
            for (int i = 0; i < n; ++i) {
                int d = data[i];
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
        if (!do_extend) updated_to = extend_to;             // completion only: the loop below does not run
        double B;
        if (leaf_status == -1) B = 0;
        else if (leaf_status == 1) B = cx.Ltree;
        else B = r_tracked_len(t, n, present_mask);

        PF_ACC_DECL;
        while (updated_to < extend_to) {
            PF_TICK(tk0);
            double new_to = extend_to < next_base ? extend_to : next_base;
            double f = fastexp(-v_mu * B * (new_to - updated_to));
            w_post *= f;
            w_pilot *= f;
            if (BIASED && cx.gK > 0) {
                // importance_weight_over_segment (particle.cpp:1138-1181): true over guide rate for the stretch
                // without recombination
                double dist = new_to - updated_to;
                double target_rate = dist * v_rho * cx.Ltree;
                double sampled_rate = dist * cx.grho[cx.ridx] * cx.Ltree;
                double iws = fastexp(sampled_rate - target_rate);
                w_post *= iws;
                w_pilot *= iws;
            }
            updated_to = new_to;
            if (BIASED && cx.gK > 0 && updated_to < extend_to && cx.ridx + 1 < cx.gK && updated_to == cx.gpos[cx.ridx + 1]) {
                // reached a change of the guide rate: no genealogy change, new draw under the new rate
                // (particle.cpp:822-826); the open stretch continues
                cx.ridx += 1;
                next_base = r_sample_next_base<false>(cx, updated_to);
                if constexpr (PIPE) r_draws_prefetch(cx);
                continue;
            }
            PF_TICK(tk1);
            PF_ACC(0, tk0, tk1);
            if (updated_to < extend_to) {
                double* rec = rec_ptr(A, p, widx);
                rec[0] = x_mark;
                rec[1] = updated_to;
#pragma unroll
                for (int r = 0; r < RTree<NM>::NI; ++r) if (r < n - 1) rec[5 + r] = t.S[r];
                double h, tc;
                PF_TICK(tk2);
                PF_ACC(1, tk1, tk2);
                r_genealogy_update<NM, BIASED, PIPE>(cx, t, &h, &tc);
                PF_TICK(tk3);
                PF_ACC(2, tk2, tk3);
                if (cx.vbc) { w_post *= cx.upd_fac; w_pilot *= cx.upd_fac; cx.upd_fac = 1.0; }
                rec[2] = h;
                rec[3] = tc;
                rec[4] = __longlong_as_double((long long)make_meta(0, mark_limit, limit, n, cx.last_desc, TREES ? cx.last_desc_new : 0u));
                ++widx;
                if (leaf_status == 0) B = r_tracked_len(t, n, present_mask);
                if (leaf_status == 1) B = cx.Ltree;
                PF_TICK(tk4);
                PF_ACC(3, tk3, tk4);
                pf_acc_trip(pf_acc);
                if (BIASED) {
                    // particle.cpp:866-891: immediate vs delayed application of the importance weight
                    double iw = cx.last_iw, rbiw = cx.last_rbiw;
                    double delay_height = (A.delay_type & 3) == 0 ? h : tc;
                    int idx = 0;
                    while (idx + 1 < cx.nb + 1 && sBH[idx + 1] < delay_height) ++idx;
                    if (idx >= cx.nb) idx = cx.nb - 1;
                    if (sBS[idx] == 1.0 && !(A.delay_type & 4)) { w_post *= rbiw; w_pilot *= rbiw; iw /= rbiw; }   // bit 2: every factor delayed (pf_model.delay_type)
                    double delay = A.app_delays[r_epoch_of(cx, delay_height)];
                    d_adjust_with_delay(ds, w_post, w_pilot, iw, delay, updated_to);
                }
                PF_TICK(tk5);
                next_base = r_sample_next_base<true, PIPE>(cx, updated_to);      // fourth uniform of the update
                x_mark = updated_to;
                mark_limit = limit;
                PF_TICK(tk6);
                PF_ACC(4, tk5, tk6);
            }
        }
        PF_ACC_STORE;

        PF_STAMP(5);
        // a fresh handle on the argument block (pf_reopen): what the rest of the row needs from it -- the pointers of the output
        // slot, the scan rings -- is loaded from here on and does not sit in scalar registers across the update loop
        const KA& A1 = pf_reopen(A);
        const DState st1 = state_slot(A1, cur);
        if (BIASED) {
            // apply the factors that fell due during this extension (particle.cpp:910-916)
            for (;;) {
                if (ds.count == 0 || !do_extend) break;
                double pm = ds.pos[0];
                for (int i = 1; i < ds.count; ++i) { double pi = ds.pos[(size_t)i * ds.Np]; if (pi < pm) pm = pi; }
                if (!(pm < extend_to)) break;
                d_apply_earliest(ds, w_pilot);
            }
            st1.dcount[p] = ds.count;
            st1.total_delayed[p] = ds.total;
            if (cx.gK > 0) st1.ridx[p] = cx.ridx;
            has_pending = ds.count > 0;
        }
        if (do_extend && A1.seg_state[s] == 0) {
            const bool dephase = A1.flags & 2;
            const bool anc = A1.flags & 1;
            unsigned het_pairs = 0;
            int ncfg = 1;
            for (int i = 0; i + 1 < n; i += 2) {
                bool d0_two = (two_mask >> i) & 1u;
                bool one0 = (one_mask >> i) & 1u, one1 = (one_mask >> (i + 1)) & 1u;
                bool zero0 = (zero_mask >> i) & 1u, zero1 = (zero_mask >> (i + 1)) & 1u;
                bool het = d0_two || (dephase && ((one0 && zero1) || (zero0 && one1)));
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
            r_site_branch_probs(t, n, v_mu, bprob);     // once per row: the phasing configurations share them
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

        PF_STAMP(6);
#pragma unroll
        for (int r = 0; r < RTree<NM>::NI; ++r)
            if (r < n - 1) {
                st1.S[(size_t)r * A1.Np + p] = t.S[r];
                st1.C[(size_t)(2 * r) * A1.Np + p] = (int8_t)t.C0[r];
                st1.C[(size_t)(2 * r + 1) * A1.Np + p] = (int8_t)t.C1[r];
            }
        st1.w_post[p] = w_post;
        st1.w_pilot[p] = w_pilot;
        st1.next_base[p] = next_base;
        st1.x_mark[p] = x_mark;
        st1.mark_limit[p] = mark_limit;
        st1.Ltree[p] = cx.Ltree;
        A1.rng_ctr[p] = cx.ctr;
        A1.ebuf[p] = cx.ebuf;
        if constexpr (PIPE) {
            if (PR.draws) A1.dt_ctr[(size_t)(PR.draws - 1) * A1.Np + p] = cx.ctr;      // where the next row's draws start
            A1.rg_widx[(size_t)cur * A1.Np + p] = widx;
            if (!do_extend) A1.widx[p] = widx;              // the state goes back to the general kernels
            if (PR.ahead) {
                // running ahead of the counts: what they check of the ring (records appended as of the row they count) cannot see
                // this row's writes, so the writer checks them itself against the oldest generation any pending count can ask for
                const unsigned kold = A1.gstart[(size_t)(A1.ctrl->g_safe % A1.Gcap) * A1.Np + p];
                if (widx - kold > A1.cap && !A1.ctrl->err) A1.ctrl->err = ERR_LOG_OVERFLOW;
            }
        } else {
            A1.widx[p] = widx;
        }
        if (TREES && widx >= A1.cap) A1.ctrl->err = ERR_LOG_OVERFLOW;       // -arg keeps every record
        if constexpr (!PIPE) {
#pragma unroll
            for (int r = 0; r < RTree<NM>::NI; ++r) if (r < n - 1) A1.snap_S[A1.sp][(size_t)r * A1.Np + p] = t.S[r];
            A1.snap_w[A1.sp][p] = w_post; A1.snap_xm[A1.sp][p] = x_mark; A1.snap_ml[A1.sp][p] = mark_limit; A1.snap_widx[A1.sp][p] = widx;
        }
    }
    PF_STAMP(7);
    const KA& A2 = pf_reopen(A);
    double sp = wave_tree_sum(w_post);
    double sq = wave_tree_sum(w_pilot * w_pilot);
    double sc = wave_hs_scan(w_pilot, lane);
    double scp = wave_hs_scan(w_post, lane);
    double scm = wave_max_scan_d(sc, lane);     // running max of the pilot scan (a parallel FP scan need not be monotone)
    long long chunk = p >> 6;
    PF_STAMP(8);
    if constexpr (PIPE) {
        const size_t ro = (size_t)cur * A2.Np, co = (size_t)cur * A2.nc;
        if (active) { A2.rg_scan1[ro + p] = sc; A2.rg_scanp[ro + p] = scp; A2.rg_scan1m[ro + p] = scm; }
        if (p == A2.Np - 1) A2.ctrl->last1[cur] = sc;
        if (lane == 63 && chunk < A2.nc) {
            A2.rg_cpost[co + chunk] = sp;
            A2.rg_csq[co + chunk] = sq;
            A2.rg_cpil[co + chunk] = sc;
            A2.rg_cpp[co + chunk] = scp;
            A2.rg_cmx1[co + chunk] = scm;
        }
        if (BIASED) {
            unsigned long long pend = __ballot(has_pending);
            if (lane == 0 && chunk < A2.nc) {
                A2.rg_dpend[co + chunk] = __popcll(pend);
                if (!PR.extend) A2.chunk_dpend[chunk] = __popcll(pend);     // the state goes back to the general kernels
            }
        }
    } else {
        if (active) { A2.scan1[p] = sc; A2.scanp2[A2.sp][p] = scp; A2.scan1m[p] = scm; }
        if (lane == 63 && chunk < (A2.Np + 63) / 64) {
            A2.chunk_post[chunk] = sp;
            A2.chunk_sq[chunk] = sq;
            A2.chunk_pil[chunk] = sc;
            A2.chunk_pp[chunk] = scp;
            A2.chunk_mx1[chunk] = scm;
        }
        if (BIASED) {
            unsigned long long pend = __ballot(has_pending);
            if (lane == 0 && chunk < (A2.Np + 63) / 64) A2.chunk_dpend[chunk] = __popcll(pend);
        }
    }
}

template <int NM, bool BIASED, bool TREES = false>
__global__ __launch_bounds__(PF_BS) void k_extend_reg(KArgs A, long long s, int fuse) {
    extend_reg_body<NM, BIASED, false, TREES>(A, s, fuse);
}

// ------------------------------------------------------------------ count bookkeeping (shared)
// Ordered reduction of the k_count partials of the previous step into the totals
// (count.cpp:407-414).  Every k_count workgroup adds its step result to its own accumulator (same workgroup, same
// address, kernels in stream order: no race), so nothing has to be folded per row; k_count_fin folds the
// accumulators into the totals when the host asks for them (pf_sync).
#define PF_FIN_TILE_B 64       // k_count workgroup partials staged per pass
#define PF_FIN_PAIRS 64        // (epoch, statistic) pairs staged per pass

template <class KA>
__device__ void finalize_counts(const KA& A, Ctrl* c, int tid, int nthreads, double* stage /* PF_FIN_PAIRS*PF_FIN_TILE_B */) {
    const int E = A.E;
    const int nb = A.nbx;
    const int NC = A.ncol;
    const int npairs = E * NC;
    // All accumulators of a tile are fetched with independent loads (a serial load-add chain would pay the
    // full memory latency per term), then each pair is summed in workgroup order from LDS.
    for (int p0 = 0; p0 < npairs; p0 += PF_FIN_PAIRS) {
        const int np_ = npairs - p0 < PF_FIN_PAIRS ? npairs - p0 : PF_FIN_PAIRS;
        double run = 0.0;                                   // running sum of pair (p0 + tid), threads < np_
        for (int b0 = 0; b0 < nb; b0 += PF_FIN_TILE_B) {
            const int nbt = nb - b0 < PF_FIN_TILE_B ? nb - b0 : PF_FIN_TILE_B;
            for (int idx = tid; idx < np_ * nbt; idx += nthreads) {
                int pp = idx / nbt, b = idx % nbt;
                int pair = p0 + pp;
                int e = pair / NC, k = pair % NC;
                double* src = &A.partial[((size_t)e * A.nbx + b0 + b) * NC + k];
                stage[pp * PF_FIN_TILE_B + b] = *src;
                *src = 0.0;
            }
            __syncthreads();
            if (tid < np_)
                for (int b = 0; b < nbt; ++b) run += stage[tid * PF_FIN_TILE_B + b];
            __syncthreads();
        }
        if (tid < np_) {
            int pair = p0 + tid;
            int e = pair / NC, k = pair % NC;
            A.totals[(size_t)k * E + e] += run;
        }
    }
    __syncthreads();
    if (tid == 0) {
        c->first_epoch = E;
        c->count_active = 0;
        c->pending_fin = 0;
    }
    __syncthreads();
}

// generation bookkeeping for the windows of the coming count step: g_lo[e] / g_hi[e] = generations
// that hold the window ends (both monotone along the sweep: amortised O(1) per step)
template <class KA>
__device__ void window_generations(const KA& A, Ctrl* c, const Windows& W, int tid) {
    const int E = A.E;
    const int G = c->gen_prev;
    if (tid < E) {
        const int e = tid;
        int g = c->g_lo[e];
        while (g < G && A.gen_x0[(g + 1) % A.Gcap] <= W.a[e]) ++g;
        c->g_lo[e] = g;
        if (e >= W.first) {
            int h = c->g_hi[e];
            if (h < g) h = g;
            while (h < G && A.gen_x0[(h + 1) % A.Gcap] < W.b[e]) ++h;
            c->g_hi[e] = h;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int gr = c->g_lo[0];
        for (int e = 1; e < E; ++e) gr = c->g_lo[e] < gr ? c->g_lo[e] : gr;
        c->g_retain = gr;
        c->delayed_opp += W.b[E - 1] - W.a[E - 1];
        if (A.n_bias > 0) {
            // update_delayed_weight_count (count.cpp:395-397): particles that carry pending factors
            long long npend = 0;
            const int ncq = (int)((A.Np + 63) / 64);
            for (int q = 0; q < ncq; ++q) npend += A.chunk_dpend[q];
            c->delayed_count += (double)npend * (W.b[E - 1] - W.a[E - 1]);
        }
        c->first_epoch = W.first;
        c->count_active = W.first < E;
        if (W.first < E) c->pending_fin = 1;
        c->nbx_used = A.nbx;
    }
}

// ------------------------------------------------------------------ k_decide
// normalize_probability (pc.cpp:420-438), the ESS test of resample (pc.cpp:247-283) and
// systematic_resampling (pc.cpp:474-504) -> offspring table, parent table, survivor run list.
//
// Grid = one workgroup per 256 particles + one bookkeeping workgroup.  A single workgroup has four
// SIMDs, far too few for anything per-particle, so:
//   * every particle workgroup redundantly redoes the tiny level-2/3 part of the canonical reduction
//     (<= 4096 per-wavefront partials): all of them obtain bit-identical T, S1, S2, ESS, flag, u without
//     any inter-workgroup wait;
//   * if resampling is due, each computes the final offspring offsets of its own particles (one per lane),
//     the parent table entries of their offspring and its survivor count: no inter-workgroup wait;
//   * the bookkeeping workgroup advances the window
//     generations (off the critical path).
template <class KA>
__device__ __forceinline__ void decide_body(const KA& A, long long s, int mode, const Windows& W, int nblocks) {
    __shared__ double l2s[4096];          // level-2 inclusive scan of the per-wavefront pilot totals / finalize staging
    __shared__ double l2_post[64], l2_sq[64], l2_tot[64], l2_totp[64];
    __shared__ int wred[PF_BS / 64];
    __shared__ double wredd[PF_BS / 64];
    __shared__ int slo[PF_BS];
    __shared__ double pm[4096];           // exclusive prefix max over chunks of the pilot prefix sums
    Ctrl* c = A.ctrl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = PF_BS / 64;
    if (pf_bx() == nblocks) {
        window_generations(A, c, W, tid);
        return;
    }
    const long long Np = A.Np;
    const int nc = (int)((Np + 63) / 64);
    const int ng = (nc + 63) / 64;
    const double last_scan1 = A.scan1[Np - 1];
    const long long i_own = (long long)pf_bx() * PF_BS + tid;
    const double own_scan1m = (i_own > 0 && i_own < Np) ? A.scan1m[i_own - 1] : 0.0;
    const double next_scan1m = (i_own + 1 < Np) ? A.scan1m[i_own] : 0.0;
    // generation / event counters as published by the previous k_resample: workgroup 0 advances the live ones
    // at the end of this kernel while other workgroups may still be starting
    const int G = c->gen_prev;
    const unsigned long long n_res = (unsigned long long)c->nres_prev;
    // level 2 of the canonical radix-64 reduction / scans
    for (int g = wave; g < ng; g += nwaves) {
        int ch = g * 64 + lane;
        double vp = ch < nc ? A.chunk_post[ch] : 0.0;
        double vs = ch < nc ? A.chunk_sq[ch] : 0.0;
        double vl = ch < nc ? A.chunk_pil[ch] : 0.0;
        double vq = ch < nc ? A.chunk_pp[ch] : 0.0;
        double rp = wave_tree_sum(vp);
        double rs = wave_tree_sum(vs);
        double sc = wave_hs_scan(vl, lane);
        double scq = wave_hs_scan(vq, lane);
        if (ch < nc) { l2s[ch] = sc; if (pf_bx() == 0) A.l2scanp[ch] = scq; }
        if (lane == 63) { l2_post[g] = rp; l2_sq[g] = rs; l2_tot[g] = sc; l2_totp[g] = scq; }
    }
    __syncthreads();
    // level 3, redundantly in every thread (at most 64 group totals): no further barrier needed
    double T, S2;
    {
        double vp = lane < ng ? l2_post[lane] : 0.0;
        double vs = lane < ng ? l2_sq[lane] : 0.0;
        T = wave_tree_sum(vp);
        S2 = wave_tree_sum(vs);
    }
    auto chunk_offset = [&](int ch) -> double {
        double run = 0.0;
        const int gq = ch / 64;
        for (int g = 0; g < gq; ++g) run = run + l2_tot[g];
        double off = (ch % 64 == 0) ? 0.0 : l2s[ch - 1];
        return run + off;
    };
    if (pf_bx() == 0) {
        for (int ch = tid; ch < nc; ch += PF_BS) {
            A.chunk_off[ch] = chunk_offset(ch);
            double runp = 0.0;
            for (int g = 0; g < ch / 64; ++g) runp = runp + l2_totp[g];
            double offp = (ch % 64 == 0) ? 0.0 : A.l2scanp[ch - 1];
            A.chunk_offp2[A.sp][ch] = runp + offp;
        }
    }
    const double S1 = chunk_offset(nc - 1) + last_scan1;   // inclusive scan at the last particle (= oracle incl[N-1])
    const double ess = (S1 * S1) / S2;
    const int flag = (mode == 0 && ess < A.ess_threshold - 1e-6) ? 1 : 0;
    double pos = A.L;
    if (mode == 0) {
        double seg_end = A.seg_start[s] + A.seg_len[s];
        pos = seg_end < A.L ? seg_end : A.L;
    }
    const double u = flag ? philox_uniform(A.seed, 0xFFFFFFFFu, 1, n_res) : 0.0;
    if (pf_bx() == 0 && tid == 0) {
        if (!(T > 0.0) && !c->err) c->err = ERR_ZERO_PROB;       // the first error is the one reported
        double logl = c->logl + dlog(T);
        c->logl = logl;
        double inv = 1.0 / T;
        if (mode == 0) {
            A.tr_T[s] = T;
            A.tr_ess[s] = ess;
            A.tr_flag[s] = flag;
            A.tr_logl[s] = logl;
            c->cur_pos = pos;
        }
        c->T = T; c->inv_T = inv; c->S1 = S1; c->S2 = S2; c->ess = ess; c->u = u; c->flag = flag;
        c->step[A.sp].inv_T = inv; c->step[A.sp].G = G; c->step[A.sp].flag = flag;
    }
    if (!flag) return;

    // ---- systematic resampling (pc.cpp:474-504) in closed form, one particle per lane, no inter-workgroup wait ----
    // lo[i] = max_{j<=i} lo_raw(incl[j-1]),  lo_raw(x) = #{ j in [0,N) : (j+u) * S1 < N * x }  (pc.cpp:491 scaled by
    // N*S1: no division).  lo_raw is monotone in x, so the running max can be taken on the prefix sums instead:
    // max_{j<=m} incl[j] = max( max_{c'<c} (chunk_off[c'] + mx1[c']),  chunk_off[c] + runmax(scan1)[m] ), which every
    // workgroup derives from the per-wavefront summaries alone.
    const double dn = (double)Np;
    const double invS1 = 1.0 / S1;
    {
        const int perc = (nc + PF_BS - 1) / PF_BS;
        const int c0 = tid * perc, c1 = c0 + perc < nc ? c0 + perc : nc;
        double run = 0.0;
        for (int ch = c0; ch < c1; ++ch) { double vch = chunk_offset(ch) + A.chunk_mx1[ch]; run = vch > run ? vch : run; }
        double scd = wave_max_scan_d(run, lane);
        if (lane == 63) wredd[wave] = scd;
        __syncthreads();
        double pre = 0.0;
        for (int w = 0; w < wave; ++w) pre = wredd[w] > pre ? wredd[w] : pre;
        double before = __shfl_up(scd, 1, 64);
        if (lane > 0) pre = before > pre ? before : pre;
        run = pre;                           // exclusive prefix for this thread's first chunk
        for (int ch = c0; ch < c1; ++ch) {
            pm[ch] = run;
            double vch = chunk_offset(ch) + A.chunk_mx1[ch];
            run = vch > run ? vch : run;
        }
        __syncthreads();
    }
    auto lo_at = [&](long long i, double s1m) -> int {      // final offspring offset of particle i (0 < i < Np)
        const int ch = (int)((i - 1) >> 6);
        double incl = chunk_offset(ch) + s1m;
        double pmx = pm[ch];
        incl = pmx > incl ? pmx : incl;
        double rhs = dn * incl;
        double guess = floor(rhs * invS1 - u);
        long long g = guess < 0 ? 0 : (guess > dn ? Np : (long long)guess);
        while (g > 0 && !((((double)(g - 1)) + u) * S1 < rhs)) --g;
        while (g < Np && ((((double)g) + u) * S1 < rhs)) ++g;
        return (int)g;
    };
    int* lo = A.lo + (size_t)(G % A.Gcap) * (Np + 1);
    int lo_i = 0, lo_n = 0;
    const bool act = i_own < Np;
    if (act) {
        lo_i = i_own > 0 ? lo_at(i_own, own_scan1m) : 0;
        lo[i_own] = lo_i;
        slo[tid] = lo_i;
    }
    __syncthreads();
    int cnt = 0;
    if (act) {
        if (i_own + 1 >= Np) { lo_n = (int)Np; lo[Np] = (int)Np; }
        else if (tid + 1 < PF_BS) lo_n = slo[tid + 1];
        else lo_n = lo_at(i_own + 1, next_scan1m);          // first particle of the next workgroup (recomputed)
        cnt = lo_n - lo_i;
        for (int q = lo_i; q < lo_n; ++q) A.parent[q] = (int)i_own;
    }
    // survivors per workgroup (k_resample turns them into the run list of the ending generation)
    unsigned long long bal = __ballot(cnt > 0);
    if (lane == 0) wred[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < nwaves; ++w) tot += wred[w];
        A.blkcnt2[A.sp][pf_bx()] = tot;
        if (pf_bx() == 0) {
            // toggle buffers / open the next generation
            int ev = (int)n_res;
            if (ev < A.max_trace_events) A.ev_seg[ev] = (int)s;
            c->cur ^= 1;
            c->gen = G + 1;
            A.gen_x0[(G + 1) % A.Gcap] = pos;
            c->n_resample = (long long)n_res + 1;
            if (G + 1 - c->g_retain >= A.Gcap - 1) c->err = ERR_GEN_OVERFLOW;
            if (A.rec_trees && G + 2 >= A.Gcap) c->err = ERR_GEN_OVERFLOW;      // -arg keeps every generation
        }
    }
}

// ------------------------------------------------------------------ k_count
// update_all_counts_single_evolevent (count.cpp:495-555) evaluated from the compact per-slot
// records.  blockIdx.y = epoch; the x dimension strides over the live ancestors (runs of the
// composite ancestor maps) of every generation whose positions intersect the epoch's window.
// per-thread statistics of one epoch: [coal count, opp, weight][P], recomb count, opp, weight and -- for
// structured models -- [migration count][P][P], [migration opp, weight][P]  (count.hpp:95-110)
template <int P>
struct AccT {
    static constexpr int NC = P == 1 ? 6 : 3 * P + 3 + P * P + 2 * P;
    static constexpr int CC = 0, CO = P, CW = 2 * P, RC = 3 * P, RO = 3 * P + 1, RW = 3 * P + 2;
    static constexpr int MC = 3 * P + 3, MO = 3 * P + 3 + P * P, MW = 3 * P + 3 + P * P + P;
    double v[NC];
};

__device__ __forceinline__ double ovl(double a0, double a1, double b0, double b1) {
    double lo = a0 > b0 ? a0 : b0;
    double hi = a1 < b1 ? a1 : b1;
    double d = hi - lo;
    return d > 0.0 ? d : 0.0;
}

// sum_i (nl - i) * |[S_{i-1}, S_i] n [max(T0,lo_t), min(T1,hi_t)]|, optionally including the single
// lineage above the top node (coalescence paths).  S lives in registers (NI compile-time).
template <int NI>
__device__ __forceinline__ double slice_len(const double (&S)[NI], int nint, int nl, double T0, double T1, double lo_t,
                                            double hi_t, bool above_top) {
    double a = T0 > lo_t ? T0 : lo_t;
    double b = T1 < hi_t ? T1 : hi_t;
    if (!(b > a)) return 0.0;
    double acc = 0.0, prev = 0.0;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (i < nint) {
            double top = S[i];
            acc += (double)(nl - i) * ovl(prev, top, a, b);
            prev = top;
        }
    if (above_top) acc += (double)(nl - nint) * ovl(prev, PF_INF, a, b);
    return acc;
}

struct Win { int e, rf; double T0, T1, a_e, b_e; bool end_seq; };

// local recombination map (record_local_recomb_events, count.cpp:559-613): per workgroup the differential
// opportunity of its window is collected in LDS bins and flushed with one global atomic per touched bin
#define PF_LBINS 2048
struct LMap {
    double* lds;          // [PF_LBINS] bins of this workgroup, bin 0 = interval b0
    int nlds;             // bins in use: the window of the step spans few of them (the rest of a step's additions, if any, go to memory)
    long long b0;
    double* gopp;         // global differential opportunity (null: map not recorded)
    double* gcnt;         // global counts [(n+2)][nbins]
    long long nbins;
    int ncnt;             // bins whose (n+2) count rows are collected in LDS too, behind the nlds opportunity bins (0: the counts go to memory)
};
__device__ __forceinline__ void lmap_add(const LMap& L, long long idx, double v) {
    if (idx < 0 || idx >= L.nbins) return;
    long long k = idx - L.b0;
    // the bins are LDS: said in the pointer's type, so that the addition is a ds_add_f64 and not a flat atomic that finds out per lane
    typedef __attribute__((address_space(3))) double lds_double;
    if (k >= 0 && k < L.nlds) __hip_atomic_fetch_add((lds_double*)L.lds + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(&L.gopp[idx], v);
}
// constant opportunity density over [x_lo, x_hi): the differential encoding of count.cpp:578-588.  When the stretch spans more than one
// interval the two additions at its left end depend on (x_lo, density) only and the two at its right end on (x_hi, density) only.
__device__ __forceinline__ void lmap_left(const LMap& L, double x_lo, double dens) {
    const double iv = 100.0;
    const long long first = (long long)(x_lo / iv);
    const double fi = (double)(first + 1) * iv - x_lo;
    lmap_add(L, first, fi * dens);
    lmap_add(L, first + 1, (iv - fi) * dens);
}
__device__ __forceinline__ void lmap_right(const LMap& L, double x_hi, double dens) {
    const double iv = 100.0;
    const long long last = (long long)(1 + x_hi / iv);
    const double li = x_hi - (double)(last - 1) * iv;
    lmap_add(L, last - 1, (li - iv) * dens);
    lmap_add(L, last, -(li * dens));
}
__device__ __forceinline__ void lmap_opportunity(const LMap& L, double x_lo, double x_hi, double weight, double opp) {
    const double iv = 100.0;
    long long first = (long long)(x_lo / iv);
    long long last = (long long)(1 + x_hi / iv);
    double dens = weight * opp / (x_hi - x_lo);
    if (first == last - 1) {
        double top = (double)(first + 1) * iv;
        double first_interval = (top < x_hi ? top : x_hi) - x_lo;
        lmap_add(L, first, first_interval * dens);
        lmap_add(L, first + 1, -(first_interval * dens));
    } else {
        // (keeping the densities of the ends that coincide with the window's -- most of them -- in two registers per lane and adding them
        // once per wavefront was measured: slower, 3.29e4 -> 3.20e4 segments/s for one chunk, the registers cost more than the conflicts)
        lmap_left(L, x_lo, dens);
        lmap_right(L, x_hi, dens);
    }
}
// a recombination event at event_base, height h, below which the samples `desc` hang (count.cpp:590-612)
__device__ __forceinline__ void lmap_event(const LMap& L, int n, double event_base, double h, unsigned desc, double weight) {
    long long idx = (long long)(event_base / 100.0);
    if (idx < 0 || idx >= L.nbins) return;
    const int nd = __popc(desc);
    const long long k = idx - L.b0;
    if (k >= 0 && k < L.ncnt) {
        // the workgroup's own rows of the counts: a row's events all fall into the few intervals of its epoch's window, and every one
        // of them used to be n + 2 additions on memory, to the same handful of addresses as everybody else's
        typedef __attribute__((address_space(3))) double lds_double;
        lds_double* base = (lds_double*)L.lds + L.nlds + k;
        for (int i = 0; i < n; ++i)
            if ((desc >> i) & 1u) __hip_atomic_fetch_add(base + (size_t)i * L.ncnt, weight / nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(base + (size_t)n * L.ncnt, weight * h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(base + (size_t)(n + 1) * L.ncnt, weight * dlog(h + 1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    for (int i = 0; i < n; ++i)
        if ((desc >> i) & 1u) atomicAdd(&L.gcnt[(size_t)i * L.nbins + idx], weight / nd);
    atomicAdd(&L.gcnt[(size_t)n * L.nbins + idx], weight * h);
    atomicAdd(&L.gcnt[(size_t)(n + 1) * L.nbins + idx], weight * dlog(h + 1.0));
}

template <int NI, int P, class KA>
__device__ __forceinline__ void stretch_contrib(AccT<P>& acc, const KA& A, const Win& W, const LMap& L, double w, double x0,
                                                double x1, const double (&S)[NI], int lim_start) {
    if (!(W.rf & REC_RECOMB) || W.e > lim_start) return;
    double xs = ovl(x0, x1, W.a_e, W.b_e);
    if (!(xs > 0.0)) return;
    double len = slice_len<NI>(S, A.n - 1, A.n, W.T0, W.T1, 0.0, PF_INF, false);
    // a tree that does not reach the epoch has no opportunity there: nothing to add to the sums (adding zero leaves them as
    // they are) and nothing to the map -- most (record, epoch) pairs of the old epochs, whose windows hold the most records
    if (!(len > 0.0)) return;
    double opp = len * xs;
    acc.v[AccT<P>::RO] += w * opp;
    acc.v[AccT<P>::RW] += w * w * opp;
    if (L.gopp) lmap_opportunity(L, x0 > W.a_e ? x0 : W.a_e, x1 < W.b_e ? x1 : W.b_e, w, opp);
}

// What one record adds to the sums of one epoch window (update_all_counts_single_evolevent, count.cpp:495-555): f0..f4 are the
// record's head [x0, x1, h, t_c | piece reference, meta], S the heights of the tree in force over [x0, x1].
template <int NI, int P, class KA>
__device__ __forceinline__ void record_contrib_one(AccT<P>& acc, const KA& A, const Win& W, const LMap& L, double w, long long a,
                                                   double f0, double f1, double f2, double f3, double f4, const double (&S)[NI]) {
    using AC = AccT<P>;
    const int n = A.n;
    double x0 = f0, x1 = f1;
    if (x1 < W.a_e) return;        // consumed by earlier windows
    unsigned long long meta = (unsigned long long)__double_as_longlong(f4);
    int type = (int)(meta & 0xff);
    int lim_start = (int)((meta >> 8) & 0xff) - 1;
    int lim_event = (int)((meta >> 16) & 0xff) - 1;
    int n_eff = (int)((meta >> 24) & 0xff);
    if (type <= 1) stretch_contrib<NI, P>(acc, A, W, L, w, x0, x1, S, lim_start);
    if (type == 0 || type == 2) {
        double h = f2;
        bool inwin = (W.a_e <= x1) && (x1 < W.b_e);
        if constexpr (P == 1) {
            double tc = f3;
            if (inwin && (W.rf & REC_COALMIGR) && W.e <= lim_event && tc >= W.T0 && h < W.T1) {      // (the floating lineage's path [h, t_c] meets the epoch)
                double opp = slice_len<NI>(S, n_eff - 1, n_eff, W.T0, W.T1, h, tc, true);
                acc.v[AC::CO] += w * opp;
                acc.v[AC::CW] += w * w * opp;
                if (W.T0 <= tc && tc < W.T1) acc.v[AC::CC] += w;
            }
        } else {
            // structured models: the walk of the update left pieces (pf_mp.h PLog) -- population, partners, [t0, t1),
            // event at t1 -- that are clipped to this epoch here; record flags and the epoch limit as above
            (void)n_eff;
            // without a tree dump the record says which epochs its pieces touch (piece_span, pf_mp.h)
            const unsigned span = (unsigned)((meta >> 48) & 0xffff);
            const bool elsewhere = !A.rec_trees && (span & 0x1000u) && (W.e < (int)(span & 63u) || W.e > (int)((span >> 6) & 63u));
            if (inwin && (W.rf & REC_COALMIGR) && W.e <= lim_event && !elsewhere) {
                unsigned long long ref = (unsigned long long)__double_as_longlong(f3);
                unsigned pstart = (unsigned)(ref & 0xffffffffu), np_ = (unsigned)(ref >> 32);
                if (A.pidx[a] - pstart > A.pcap || np_ > A.pcap) { if (!A.ctrl->err) A.ctrl->err = ERR_LOG_OVERFLOW; np_ = 0; }
                for (unsigned j = 0; j < np_; ++j) {
                    const double* q = A.plog + ((size_t)a * A.pcap + ((pstart + j) % A.pcap)) * 3;
                    long long tag = __double_as_longlong(q[0]);
                    const double t0 = q[1], t1 = q[2];
                    const int pop = (int)(tag & 0xff), kind = (int)((tag >> 8) & 0xff), to = (int)((tag >> 16) & 0xff);
                    const double lo = t0 > W.T0 ? t0 : W.T0, hi = t1 < W.T1 ? t1 : W.T1;
                    const double mo = hi > lo ? hi - lo : 0.0;
                    const bool ev = (kind & 3) != 0 && W.T0 <= t1 && t1 < W.T1;
                    if (!(mo > 0.0) && !ev) continue;
                    const double co = (double)((tag >> 24) & 0xff) * mo;
#pragma unroll
                    for (int pp = 0; pp < P; ++pp)
                        if (pp == pop) {
                            acc.v[AC::CO + pp] += w * co;
                            acc.v[AC::CW + pp] += w * w * co;
                            acc.v[AC::MO + pp] += w * mo;
                            acc.v[AC::MW + pp] += w * w * mo;
                            if (ev && (kind & 1)) acc.v[AC::CC + pp] += w;
                            if (ev && (kind & 2)) {
#pragma unroll
                                for (int qq = 0; qq < P; ++qq)
                                    if (qq == to) acc.v[AC::MC + pp * P + qq] += w;
                            }
                        }
                }
            }
        }
        if (type == 0) {
            bool inwin_r = (W.a_e <= x1) && ((x1 < W.b_e) || W.end_seq);
            if (inwin_r && (W.rf & REC_RECOMB) && W.e <= lim_event && W.T0 <= h && h < W.T1) {
                acc.v[AC::RC] += w;
                if (L.gopp) lmap_event(L, n, x1, h, (unsigned)((meta >> 32) & 0xffff), w);
            }
        }
    }
}

// All fields of a record are fetched in one round of independent loads before anything is tested:
// the kernel is bound by dependent-load latency, not by bytes.
template <int NI, int P, class KA>
__device__ __forceinline__ void records_contrib(AccT<P>& acc, const KA& A, const Win& W, const LMap& L, double w, long long a,
                                                unsigned k0, unsigned k1) {
    const int n = A.n;
    // A slot appends its records in the order of their positions.  Those that end before the window were consumed by earlier windows
    // (record_contrib_one's first test), those that start at or behind its end add nothing yet: on a row that lies thirty rows behind
    // the last resampling a live particle has twenty records of which the window of an old epoch meets one, and the lane with the most
    // records is what its wavefront waits for.  So the first record that can matter is looked for with three probes at a time
    // (independent loads: a round trip narrows the range to a quarter), and the walk stops at the first record past the window.
    unsigned lo = k0, len = k1 - k0;
    while (len > 4) {
        const unsigned q = len >> 2;
        const double xa = rec_ptr(A, a, lo + q)[1], xb = rec_ptr(A, a, lo + 2 * q)[1], xc = rec_ptr(A, a, lo + 3 * q)[1];
        // (x1 does not decrease along the records: the first one with x1 >= a_e lies behind every probe that is still below a_e)
        if (xc < W.a_e) { lo += 3 * q + 1; len -= 3 * q + 1; }
        else if (xb < W.a_e) { lo += 2 * q + 1; len = q; }
        else if (xa < W.a_e) { lo += q + 1; len = q; }
        else len = q;
    }
    for (unsigned k = lo; k != k1; ++k) {
        const double* rec = rec_ptr(A, a, k);
        double f0 = rec[0], f1 = rec[1], f2 = rec[2], f3 = rec[3], f4 = rec[4];
        double S[NI];
#pragma unroll
        for (int r = 0; r < NI; ++r) S[r] = r < n - 1 ? rec[5 + r] : 0.0;
        if (f0 >= W.b_e && !W.end_seq) break;           // this record and all behind it start past the window
        record_contrib_one<NI, P>(acc, A, W, L, w, a, f0, f1, f2, f3, f4, S);
    }
}

// Where a count step finds the particles of its row and the ancestor maps in force.  The general kernels keep these by
// step parity (snapshots written by k_extend) and update the run lists in place; the single-launch pipeline reads the
// row's slot of the state ring and the run-list copy that was in force for the row.
struct RunLists { int* st; int* anc; int* nruns; };
struct CountSrc {
    const double* w; const double* S; const double* xm; const int* ml; const unsigned* widx;   // the row's live particles
    const unsigned* widx_live;                    // records appended per slot as of now (ring-overwrite check)
    const double* scanp; const double* offp;      // posterior scan inside the wavefronts, exclusive offsets of the wavefronts
    RunLists lists;
    double inv; int G, g_lo, g_hi;                // normalisation of the row, its generation, generations of the epoch's window
};
template <class KA>
__device__ __forceinline__ RunLists run_lists(const KA& A, int ver) {
    RunLists r;
    r.st = ver ? A.run_st2 : A.run_st; r.anc = ver ? A.run_anc2 : A.run_anc; r.nruns = ver ? A.nruns2 : A.nruns;
    return r;
}
template <class KA>
__device__ __forceinline__ CountSrc count_src_parity(const KA& A, int sp, int e) {
    const Ctrl* c = A.ctrl;
    CountSrc q;
    q.w = A.snap_w[sp]; q.S = A.snap_S[sp]; q.xm = A.snap_xm[sp]; q.ml = A.snap_ml[sp]; q.widx = A.snap_widx[sp];
    q.widx_live = A.widx;
    q.scanp = A.scanp2[sp]; q.offp = A.chunk_offp2[sp];
    q.lists = run_lists(A, c->lver);
    q.inv = c->step[sp].inv_T; q.G = c->step[sp].G;
    q.g_lo = e < A.E ? c->g_lo[e] : 0; q.g_hi = e < A.E ? c->g_hi[e] : 0;
    return q;
}

// LDS bins of the local recombination map: one array for whichever count body the workgroup runs
__device__ __forceinline__ double* count_bins() {
    __shared__ double bins[PF_LBINS];
    return bins;
}

// LDS of a count role: one buffer per body (only the launch of the split arrangement carries both bodies)
#define PF_COUNT_LDS_BYTES 14336
template <int BYTES>
__device__ __forceinline__ char* count_lds() {
    __shared__ __attribute__((aligned(16))) char buf[BYTES];
    return buf;
}

#define PF_CNT_TILE 2048      // generations whose run counts are staged in LDS at a time
#define PF_CNT_WIDE 128       // run lists longer than this are strided over by the whole grid column
#define PF_CNT_BATCH 2        // count tasks whose first two rounds of loads a thread has in flight together

template <int NI, int P, class KA>
__device__ __forceinline__ void count_run(AccT<P>& acc, const KA& A, const CountSrc& Q, const Win& W, const LMap& L, int g, long long i,
                                          int nr, const int* rst, const int* ran, double inv) {
    const long long Np = A.Np;
    int q0 = rst[i];
    int q1 = i + 1 < nr ? rst[i + 1] : (int)Np;
    long long a = ran[i];
    // a ledger ring that has overflowed (reported by k_decide, ERR_GEN_OVERFLOW) aliases generations: whatever is
    // read then must at least stay inside the arrays and every loop must stay bounded until the host sees the error
    if (q1 <= q0 || q1 > (int)Np || q0 < 0 || a < 0 || a >= Np) return;
    // posterior mass of the descendants of (g, a): difference of the inclusive posterior scan
    const double* offp = Q.offp;
    const double* scp_ = Q.scanp;
    double hi = offp[(q1 - 1) >> 6] + scp_[q1 - 1];
    double lo = q0 > 0 ? offp[(q0 - 1) >> 6] + scp_[q0 - 1] : 0.0;
    double w = (hi - lo) * inv;
    if (!(w > 0.0)) return;
    unsigned k0 = A.gstart[(size_t)(g % A.Gcap) * Np + a];
    unsigned k1 = A.gstart[(size_t)((g + 1) % A.Gcap) * Np + a];
    if (Q.widx_live[a] - k0 > A.cap || k1 - k0 > A.cap) { if (!A.ctrl->err) A.ctrl->err = ERR_LOG_OVERFLOW; return; }
    records_contrib<NI, P>(acc, A, W, L, w, a, k0, k1);
}

// LDS of the flattened record walk of count_body: the weight, slot and first record of the task each thread found in a trip, and the
// exclusive prefix sum of the tasks' record counts
struct FlatTask { double w; int a; unsigned k0; };
__device__ __forceinline__ FlatTask* count_flat_tasks() {
    __shared__ FlatTask ft[PF_BS];
    return ft;
}
__device__ __forceinline__ int* count_flat_prefix() {
    __shared__ int pre[PF_BS + 1];
    return pre;
}

// The body of k_count for the workgroup (bx, by) of a column of nbxg workgroups; `sp` = parity of the step whose
// weights and snapshot it reads.  Called from k_count and from the count workgroups of k_row.
template <int NM, int P, bool EXACT = false, class KA>
__device__ __forceinline__ void count_body(const KA& A, const CountSrc& Q, int e, double win_a, double win_b, int bx, int nbxg) {
    constexpr int NI = NM - 1;
    using AC = AccT<P>;
    constexpr int LDS_BYTES = (int)(((sizeof(AC) * (PF_BS / 64) + 15) & ~(size_t)15) + sizeof(int) * (PF_CNT_TILE + 1 + PF_BS / 64) + 16);
    char* const cl = count_lds<LDS_BYTES>();
    AC* const red = (AC*)cl;
    int* const s_off = (int*)(cl + ((sizeof(AC) * (PF_BS / 64) + 15) & ~(size_t)15));
    int* const s_wsum = s_off + PF_CNT_TILE + 1;
    const long long Np = A.Np;
    const int n = EXACT ? NM : A.n;
    const int G = Q.G;                                  // the generation the weights belong to
    const double inv = Q.inv;
    Win W;
    W.e = e; W.rf = A.recflags[e];
    W.T0 = A.T[e]; W.T1 = e + 1 < A.E ? A.T[e + 1] : PF_INF;
    W.a_e = win_a; W.b_e = win_b;
    W.end_seq = (A.L == W.b_e);
    double* const s_lbins = count_bins();
    LMap L;
    L.lds = s_lbins; L.b0 = (long long)(W.a_e / 100.0); L.gopp = A.lmap_opp; L.gcnt = A.lmap_cnt; L.nbins = A.lmap_bins;
    {
        const long long span = (long long)(W.b_e / 100.0) - L.b0 + 4;
        L.nlds = span < 1 ? 1 : (span > PF_LBINS ? PF_LBINS : (int)span);
        L.ncnt = (A.n + 3) * L.nlds <= PF_LBINS ? L.nlds : 0;       // room for the n + 2 count rows of the same intervals
    }
    bool bins_ready = false;
    const int g_lo = Q.g_lo, g_hi = Q.g_hi;
    if (g_hi < g_lo) return;                                  // an empty window: nothing to add to this workgroup's accumulators
    const int lane = threadIdx.x & 63;
    const long long gtid = (long long)bx * PF_BS + threadIdx.x;
    const long long nthreads = (long long)nbxg * PF_BS;
    AC acc;
#pragma unroll
    for (int k = 0; k < AC::NC; ++k) acc.v[k] = 0.0;
    // Work = all (generation, run) pairs of the window plus the live particles, flattened into one task index
    // space with an LDS prefix sum of the run counts: every thread takes tasks gtid, gtid+nthreads, ... so deep
    // windows are spread over the grid column instead of being walked generation by generation.
    for (int tile_hi = g_hi; tile_hi >= g_lo; tile_hi -= PF_CNT_TILE) {
        const int tile_lo = tile_hi - PF_CNT_TILE + 1 > g_lo ? tile_hi - PF_CNT_TILE + 1 : g_lo;
        const int ntile = tile_hi - tile_lo + 1;
        __syncthreads();
        // stage run counts (live generation: Np tasks) and their exclusive prefix sum
        constexpr int PER = PF_CNT_TILE / PF_BS;
        int loc[PER];
        int lsum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            int idx = threadIdx.x * PER + k;
            int v = 0;
            if (idx < ntile) {
                int g = tile_hi - idx;
                v = g == G ? (int)Np : Q.lists.nruns[g % A.Gcap];
            }
            loc[k] = v;
            lsum += v;
        }
        int incl = lsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_wsum[w];
        int run = base + incl - lsum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            int idx = threadIdx.x * PER + k;
            if (idx <= ntile) s_off[idx] = run;
            run += loc[k];
        }
        if (threadIdx.x == PF_BS - 1) s_off[ntile] = run;      // total (idx == ntile is only reached when ntile == TILE)
        __syncthreads();
        const long long T = s_off[ntile];
        // PF_CNT_BATCH tasks per thread and trip: a task is three dependent rounds of loads (run list -> posterior scan and
        // record range -> records) and little arithmetic, and the count workgroups hide latency badly (three per CU), so
        // the first two rounds of four tasks are in flight together.  The contributions are still added in task order
        // (t, t + nthreads, ...): the sums do not change.
        // A column is launched with as many workgroups as its busiest epochs need; the window of a younger epoch sits far behind
        // the front, where few ancestors are left: most of its workgroups have no task at all and leave here, before the bins,
        // the barriers and the reduction (adding zero to their accumulators is leaving them alone).
        if (tile_hi == g_hi && tile_lo == g_lo && (long long)bx * PF_BS >= T) return;
        if (A.flags & (1 << 20)) return;          // probe: what the workgroups cost before their first task
        if (L.gopp && !bins_ready) {
            for (int k = threadIdx.x; k < L.nlds + (A.n + 2) * L.ncnt; k += PF_BS) s_lbins[k] = 0.0;
            __syncthreads();
            bins_ready = true;
        }
        // A trip of the workgroup: one task per thread -- its weight and its records found in two rounds of loads -- then the RECORDS of
        // the 256 tasks dealt out over the threads, one record per thread and pass.  A task's records used to be walked by the thread
        // that owned the task, a load and a wait each, and a wavefront lasted as long as its lane with the most records: on the rows
        // where many epochs catch up the count role ended ten microseconds after the extend role (profiles/round4/wg_trace.md).  The
        // sums are grouped differently than before round 4 (by the thread that evaluates a record, not by the task's owner), in the
        // same way in every run.
        FlatTask* const s_ft = count_flat_tasks();
        int* const s_fpre = count_flat_prefix();
        for (long long base = (long long)bx * PF_BS; base < T; base += nthreads) {          // (the same trips for every thread: barriers inside)
            const long long t = base + threadIdx.x;
            const bool tval = t < T;
            // generation of task t: last idx with s_off[idx] <= t
            int lo_i = 0, hi_i = ntile;
            while (tval && hi_i - lo_i > 1) {
                int mid = (lo_i + hi_i) >> 1;
                if ((long long)s_off[mid] <= t) lo_i = mid; else hi_i = mid;
            }
            const int tg = tile_hi - lo_i;
            const long long ti = t - s_off[lo_i];
            const int tnr = s_off[lo_i + 1] - s_off[lo_i];
            const bool tlive = tval && tg == G;
            // round 1: the run (slots [q0, q1) of the row descend from slot a of generation g)
            int q0 = 0, q1 = 0;
            long long aa = -1;
            if (tval && !tlive) {
                const int* rst = Q.lists.st + (size_t)(tg % A.Gcap) * Np;
                const int* ran = Q.lists.anc + (size_t)(tg % A.Gcap) * Np;
                q0 = rst[ti];
                q1 = ti + 1 < tnr ? rst[ti + 1] : (int)Np;
                aa = ran[ti];
            }
            // round 2: posterior mass of the run's slots (difference of the inclusive posterior scan) and the ancestor's records of that
            // generation; for a live particle its own weight, its open stretch and the records it wrote this generation.  A ledger ring
            // that has overflowed (reported by the bookkeeping, ERR_GEN_OVERFLOW) aliases generations: whatever is read then must stay
            // inside the arrays until the host sees the error.
            double w = 0.0;
            long long a_t = 0;
            unsigned k0 = 0, nrec = 0;
            if (tlive) {
                const long long a = ti;
                w = Q.w[a] * inv;
                double S[NI];
#pragma unroll
                for (int r = 0; r < NI; ++r) S[r] = r < n - 1 ? Q.S[(size_t)r * Np + a] : 0.0;
                const double xm = Q.xm[a];
                const int ml = Q.ml[a];
                const unsigned g0 = A.gstart[(size_t)(tg % A.Gcap) * Np + a];
                const unsigned g1 = Q.widx[a];
                if (!(A.flags & (1 << 21)) && w != 0.0) {
                    stretch_contrib<NI, P>(acc, A, W, L, w, xm, PF_INF, S, ml);
                    if (g1 - g0 > A.cap) { if (!A.ctrl->err) A.ctrl->err = ERR_LOG_OVERFLOW; }
                    else { a_t = a; k0 = g0; nrec = g1 - g0; }
                }
            } else if (tval && !(q1 <= q0 || q1 > (int)Np || q0 < 0 || aa < 0 || aa >= Np)) {
                const double mhi = Q.offp[(q1 - 1) >> 6] + Q.scanp[q1 - 1];
                const double mlo = q0 > 0 ? Q.offp[(q0 - 1) >> 6] + Q.scanp[q0 - 1] : 0.0;
                const unsigned rk0 = A.gstart[(size_t)(tg % A.Gcap) * Np + aa];
                const unsigned rk1 = A.gstart[(size_t)((tg + 1) % A.Gcap) * Np + aa];
                const unsigned rwl = Q.widx_live[aa];
                w = (mhi - mlo) * inv;
                if (!(A.flags & (1 << 21)) && w > 0.0) {
                    if (rwl - rk0 > A.cap || rk1 - rk0 > A.cap) { if (!A.ctrl->err) A.ctrl->err = ERR_LOG_OVERFLOW; }
                    else { a_t = aa; k0 = rk0; nrec = rk1 - rk0; }
                }
            }
            // the records of the trip's tasks, numbered through: exclusive prefix sum of the record counts over the workgroup
            int incl_r = (int)nrec;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl_r, d, 64);
                if (lane >= d) incl_r += o;
            }
            if (lane == 63) s_wsum[threadIdx.x >> 6] = incl_r;
            __syncthreads();
            int base_r = 0;
            for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) base_r += s_wsum[wv];
            s_fpre[threadIdx.x] = base_r + incl_r - (int)nrec;
            if (threadIdx.x == PF_BS - 1) s_fpre[PF_BS] = base_r + incl_r;
            s_ft[threadIdx.x].w = w; s_ft[threadIdx.x].a = (int)a_t; s_ft[threadIdx.x].k0 = k0;
            __syncthreads();
            const int nrec_all = s_fpre[PF_BS];
            for (int r = (int)threadIdx.x; r < nrec_all; r += PF_BS) {
                int lo_j = 0, hi_j = PF_BS;                  // the task of record r: last j with s_fpre[j] <= r (its own count is not zero)
                while (hi_j - lo_j > 1) { const int mid = (lo_j + hi_j) >> 1; if (s_fpre[mid] <= r) lo_j = mid; else hi_j = mid; }
                const double w_r = s_ft[lo_j].w;
                const long long a_r = s_ft[lo_j].a;
                const double* rec = rec_ptr(A, a_r, s_ft[lo_j].k0 + (unsigned)(r - s_fpre[lo_j]));
                double f0 = rec[0], f1 = rec[1], f2 = rec[2], f3 = rec[3], f4 = rec[4];
                double S[NI];
#pragma unroll
                for (int q = 0; q < NI; ++q) S[q] = q < n - 1 ? rec[5 + q] : 0.0;
                record_contrib_one<NI, P>(acc, A, W, L, w_r, a_r, f0, f1, f2, f3, f4, S);
            }
            __syncthreads();                                 // (the next trip, or the next tile's staging, writes the arrays again)
        }
    }
    // deterministic workgroup reduction: butterfly per wavefront, then wavefronts in order
    if (L.gopp && bins_ready) {
        __syncthreads();
        for (int k = threadIdx.x; k < L.nlds; k += PF_BS) {
            double v = s_lbins[k];
            long long idx = L.b0 + k;
            if (v != 0.0 && idx < L.nbins) atomicAdd(&L.gopp[idx], v);
        }
        for (int k = threadIdx.x; k < (A.n + 2) * L.ncnt; k += PF_BS) {
            const double v = s_lbins[L.nlds + k];
            const long long idx = L.b0 + k % L.ncnt;
            if (v != 0.0 && idx < L.nbins) atomicAdd(&L.gcnt[(size_t)(k / L.ncnt) * L.nbins + idx], v);
        }
    }
#pragma unroll
    for (int k = 0; k < AC::NC; ++k) acc.v[k] = wave_tree_sum(acc.v[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < AC::NC) {
        const int k = threadIdx.x;
        double t = red[0].v[k];
        for (int w = 1; w < PF_BS / 64; ++w) t += red[w].v[k];
        A.partial[((size_t)e * A.nbx + bx) * AC::NC + k] += t;      // this workgroup's own accumulator (folded by k_count_fin)
    }
}

// (Round 4 also built the counting by generation -- units of 256 tasks of one generation, their weights and record ranges fetched once
// for all the epochs whose windows meet that generation, records flattened over the workgroup: count_units_body, commit 9818bcc and
// before.  It was slower in every regime (profiles/round4/count_rebuild.md); the part of it that paid, the flattened record walk, is in
// count_body now, and the rest was taken out.)

template <int NM, int P>
__global__ __launch_bounds__(PF_BS) void k_count(KArgs A, int e0, Windows Wn) {
    const int e = e0 + (int)blockIdx.y;
    if (e < Wn.first || e >= A.E) return;
    count_body<NM, P>(A, count_src_parity(A, A.sp, e), e, Wn.a[e], Wn.b[e], (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------ k_row
// One launch per row on a single stream: workgroups [0, nb) extend the particles over row s (and complete row s-1 on
// load), the remaining workgroups evaluate the lagged counts of row s-1 (k_count's body).  The two halves touch
// disjoint data -- everything the counts read from row s-1 is immutable or double-buffered by row parity -- so the
// counting costs no synchronisation at all: no second stream, no event record / wait packets on the critical path.
template <int NM, bool BIASED, bool EXACT, bool TREES = false>
__global__ __launch_bounds__(PF_BS) void k_row(KArgs A, long long s, int fuse, int nb, int count_first, Windows Wprev) {
    if ((int)blockIdx.x < nb) {
        extend_reg_body<NM, BIASED, EXACT, TREES>(A, s, fuse);
    } else {
        const int idx = (int)blockIdx.x - nb;
        const int e = count_first + idx / nb;
        if (e < Wprev.first || e >= A.E) return;
        count_body<NM, 1, EXACT>(A, count_src_parity(A, A.sp ^ 1, e), e, Wprev.a[e], Wprev.b[e], idx % nb, nb);
    }
}

__global__ void k_count_fin(KArgs A) {
    __shared__ double stage[PF_FIN_PAIRS * PF_FIN_TILE_B];
    Ctrl* c = A.ctrl;
    if (c->pending_fin) finalize_counts(A, c, threadIdx.x, blockDim.x, stage);
}

// ------------------------------------------------------------------ k_resample (+ ledger blocks)
// Blocks [0, nblocks): no resampling -> normalise in place (pc.cpp:435-437); resampling ->
// implement_resampling (pc.cpp:321-392) as a gather from the old buffer into the new one.
// Blocks [nblocks, gridDim.x): after a resampling, re-base the run-length encoded composite
// ancestor maps of all retained generations onto the new slots (st' = lo[st], empty runs dropped).
#define PF_LEDGER_PER 8       // rounds per tile: a tile holds PF_LEDGER_PER * blockDim.x runs
#define PF_LEDGER_MAXT 1024   // largest workgroup the ledger code is launched with
#define PF_LEDGER_NEW 64      // the newest generations (long run lists) are re-based by a whole workgroup each

template <class KA>
__device__ __forceinline__ void ledger_update(const KA& A, int lb, int nlb, int G, int g_ret, RunLists src, RunLists dst) {
    __shared__ int scnt[PF_LEDGER_PER * (PF_LEDGER_MAXT / 64)], sexc[PF_LEDGER_PER * (PF_LEDGER_MAXT / 64)];
    __shared__ int stot;
    const long long Np = A.Np;
    const int* lo = A.lo + (size_t)(G % A.Gcap) * (Np + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = (int)blockDim.x, NW = NT >> 6;
    const int NCNT = PF_LEDGER_PER * NW;                 // <= 128: scanned by one wavefront, two counts per lane
    // ---- pass 1: workgroup per generation, newest PF_LEDGER_NEW generations (G itself: written by k_decide) ----
    // Compaction of a tile of PF_LEDGER_PER * blockDim.x runs with coalesced traffic only: element i of the tile
    // belongs to (round j, wavefront, lane) = (i / NT, (i % NT) / 64, i % 64).  All starts and ancestors of the tile
    // are requested at once, then all offspring offsets: two memory round trips per tile, and with the 1024
    // threads of k_decide_ledger the newest list (the survivors of the last resampling) is a single tile.  The
    // survivors of each (round, wavefront) are counted with a ballot, the counts are scanned by the first
    // wavefront, and every lane writes its survivors to their final places.  In place is safe: every input of the
    // tile is in registers before the first barrier, and the output never passes the tile's first input.
    for (int k = 1 + lb; k < PF_LEDGER_NEW; k += nlb) {
        const int g = G - k;
        if (g < g_ret) break;
        const int* rst = src.st + (size_t)(g % A.Gcap) * Np;
        const int* ran = src.anc + (size_t)(g % A.Gcap) * Np;
        int* wst = dst.st + (size_t)(g % A.Gcap) * Np;
        int* wan = dst.anc + (size_t)(g % A.Gcap) * Np;
        const int nr = src.nruns[g % A.Gcap];
        int out_base = 0;
        for (int tile0 = 0; tile0 < nr; tile0 += PF_LEDGER_PER * NT) {
            int ns[PF_LEDGER_PER], an[PF_LEDGER_PER], nx[PF_LEDGER_PER];
#pragma unroll
            for (int j = 0; j < PF_LEDGER_PER; ++j) {
                const int i = tile0 + j * NT + tid;
                ns[j] = i < nr ? rst[i] : 0;
                an[j] = i < nr ? ran[i] : 0;
                nx[j] = (lane == 63 && i < nr) ? (i + 1 < nr ? rst[i + 1] : (int)Np) : 0;     // successor of the last lane
            }
#pragma unroll
            for (int j = 0; j < PF_LEDGER_PER; ++j) {
                const int i = tile0 + j * NT + tid;
                ns[j] = i < nr ? lo[ns[j]] : 0;
                nx[j] = (lane == 63 && i < nr) ? lo[nx[j]] : 0;
            }
            unsigned keep = 0;
#pragma unroll
            for (int j = 0; j < PF_LEDGER_PER; ++j) {
                const int i = tile0 + j * NT + tid;
                int ne = __shfl_down(ns[j], 1, 64);
                if (lane == 63) ne = nx[j];
                else if (i + 1 >= nr) ne = (int)Np;                           // successor of the list's last run: lo[Np] = Np
                const bool kp = i < nr && ne > ns[j];
                const unsigned long long bal = __ballot(kp);
                if (kp) keep |= 1u << j;
                if (lane == 0) scnt[j * NW + wave] = __popcll(bal);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ancestors too have arrived before anyone stores
            __syncthreads();
            if (tid < 64) {
                const int c0 = 2 * lane < NCNT ? scnt[2 * lane] : 0;
                const int c1 = 2 * lane + 1 < NCNT ? scnt[2 * lane + 1] : 0;
                int incl = c0 + c1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    int o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                const int excl = incl - (c0 + c1);
                if (2 * lane < NCNT) sexc[2 * lane] = excl;
                if (2 * lane + 1 < NCNT) sexc[2 * lane + 1] = excl + c0;
                if (lane == 63) stot = incl;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PF_LEDGER_PER; ++j) {
                const bool kp = (keep >> j) & 1u;
                const unsigned long long bal = __ballot(kp);
                if (kp) {
                    const int pos = out_base + sexc[j * NW + wave] + __popcll(bal & ((1ULL << lane) - 1ULL));
                    wst[pos] = ns[j];
                    wan[pos] = an[j];
                }
            }
            out_base += stot;
            __syncthreads();                 // the count arrays are reused by the next tile
        }
        if (tid == 0) dst.nruns[g % A.Gcap] = out_base;
    }
    // ---- pass 2: wavefront per generation for everything older (short lists, no workgroup barriers) ----
    const int wl = lb * NW + wave;
    const int nwl = nlb * NW;
    for (int g = G - PF_LEDGER_NEW - wl; g >= g_ret; g -= nwl) {
        const int* rst = src.st + (size_t)(g % A.Gcap) * Np;
        const int* ran = src.anc + (size_t)(g % A.Gcap) * Np;
        int* wst = dst.st + (size_t)(g % A.Gcap) * Np;
        int* wan = dst.anc + (size_t)(g % A.Gcap) * Np;
        const int nr = src.nruns[g % A.Gcap];
        int base = 0;
        for (int off = 0; off < nr; off += 64) {
            int i = off + lane;
            bool valid = i < nr;
            int stv = 0, nxt = 0, anc = 0;
            if (valid) {
                stv = rst[i];
                nxt = (i + 1 < nr) ? rst[i + 1] : (int)Np;
                anc = ran[i];
            }
            int ns = valid ? lo[stv] : 0;
            int ne = valid ? lo[nxt] : 0;
            bool keep = valid && ne > ns;
            unsigned long long bal = __ballot(keep);
            int pos = __popcll(bal & ((1ULL << lane) - 1ULL));
            // all 64 lanes loaded their inputs before anyone stores (stores depend on the ballot)
            if (keep) { wst[base + pos] = ns; wan[base + pos] = anc; }
            base += __popcll(bal);
        }
        if (lane == 0) dst.nruns[g % A.Gcap] = base;
    }
}

// ------------------------------------------------------------------ k_ledger (counting stream)
// After a resampling: workgroups [0, nblocks) write the run list of the generation that ended (its survivors,
// in slot order: start = lo[a], ancestor = a); the others re-base the run-length encoded composite ancestor
// maps of all retained generations onto the new slots (st' = lo[st], empty runs dropped).  Only k_count reads
// these lists, so the whole maintenance lives on the counting stream, off the filter's critical path.
// body of k_ledger for workgroup bx of nbt; `sp` = parity of the step whose resampling it follows up
// run list of the generation Gx that ends with a resampling: its survivors, in slot order (start = lo[a], ancestor = a).
// Position = survivors in earlier workgroups (blkcnt, counted where the offspring table was made) + rank inside this one.
template <class KA>
__device__ __forceinline__ void ledger_new_list(const KA& A, RunLists dst, const int* blkcnt, int nblocks, int bx, int Gx, int gran = 1) {
    const long long Np = A.Np;
    const long long i = (long long)bx * PF_BS + threadIdx.x;
    __shared__ int wsum[PF_BS / 64];
    const int* lox = A.lo + (size_t)(Gx % A.Gcap) * (Np + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int l0 = 0, l1 = 0;
    if (i < Np) { l0 = lox[i]; l1 = lox[i + 1]; }
    const bool surv = l1 > l0;
    unsigned long long bal = __ballot(surv);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __shared__ int sblk[1024];
    // survivors per block of 256 particles; the structured extend workgroups own 64 particles each and leave four entries per block
    for (int b = threadIdx.x; b < nblocks; b += PF_BS) {
        int v = 0;
        for (int j = 0; j < gran; ++j) v += blkcnt[b * gran + j];
        sblk[b] = v;
    }
    __syncthreads();
    int base = 0;
    for (int b = 0; b < bx; ++b) base += sblk[b];
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (surv) {
        int pos = base + __popcll(bal & ((1ULL << lane) - 1ULL));
        dst.st[(size_t)(Gx % A.Gcap) * Np + pos] = l0;
        dst.anc[(size_t)(Gx % A.Gcap) * Np + pos] = (int)i;
    }
    if (bx == nblocks - 1 && threadIdx.x == PF_BS - 1) {
        int tot = base + __popcll(bal);           // base already holds the earlier wavefronts of this workgroup
        dst.nruns[Gx % A.Gcap] = tot;
    }
}

template <class KA>
__device__ __forceinline__ void ledger_body(const KA& A, int sp, int nblocks, int bx, int nbt) {
    const Ctrl* c = A.ctrl;
    if (!c->step[sp].flag) return;
    const int Gx = c->step[sp].G;
    const RunLists lists = run_lists(A, c->lver);       // the general kernels re-base in place
    if (bx >= nblocks) {
        ledger_update(A, bx - nblocks, nbt - nblocks, Gx, c->g_retain, lists, lists);
        return;
    }
    ledger_new_list(A, lists, A.blkcnt2[sp], nblocks, bx, Gx);
}

// ------------------------------------------------------------------ k_pipe: one launch per row
// The single-launch pipeline of the register-tree kernels (one population, n <= 8, no look-ahead).  Launch s carries
//   X(s)    workgroups [0, nb): extend the particles over row s.  On load they complete row s-1 themselves: decision
//           (decide_row), offspring offsets, parent search, gather or normalisation -- nothing from k_decide is read;
//   B(s-1)  one workgroup: the bookkeeping of row s-1 (log-likelihood, traces, generations of the count windows,
//           posterior-scan offsets), published in Ctrl::ri[(s-1) & (PF_RING-1)];
//   L(s-2)  ledger upkeep after the resampling of row s-2: reads the run lists in force, writes the other copy;
//   C(s-2)  the lagged counts of row s-2 from the lists in force for that row and its slot of the state ring.
// What a role reads was written by an earlier launch (stream order) or is private to it; nothing waits inside the
// launch.  The decide kernel and its two dependent kernel boundaries per row are gone from the critical path.
__global__ void k_pipe_seed(KArgs A, int slot) {
    Ctrl* c = A.ctrl;
    Ctrl::RowInfo& r = c->ri[slot];
    r.n_res = c->n_resample; r.gen = c->gen; r.flag = 0; r.lver = c->lver; r.g_retain = c->g_retain; r.first = A.E;
    r.inv_T = 1.0; r.T = 1.0; r.S1 = 0.0; r.u = 0.0; r.pos = c->cur_pos;
    c->xr[slot].n_res = c->n_resample; c->xr[slot].gen = c->gen; c->xr[slot].flag = 0;
}

template <bool BIASED, class KA>
__device__ __forceinline__ void pipe_bookkeeping(const KA& A, const PipeLds& q, const PipeLaunch& PL, const Windows& W) {
    Ctrl* c = A.ctrl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = PF_BS / 64;
    const int E = A.E, nc = A.nc, ng = (nc + 63) / 64;
    const int slot = PL.b_slot;
    const long long n_res = c->n_resample;        // as the bookkeeping of the previous row left them (previous launch)
    const int G = c->gen;
    RowDecision d = decide_row<false>(A, q, slot, n_res);
    __syncthreads();
    // exclusive offsets of the posterior scan of the row's wavefronts: the counts turn differences of that scan into
    // the posterior mass of a particle's descendants
    const double* cpp = A.rg_cpp + (size_t)slot * nc;
    for (int g = wave; g < ng; g += nwaves) {
        int ch = g * 64 + lane;
        double vq = ch < nc ? cpp[ch] : 0.0;
        double scq = wave_hs_scan(vq, lane);
        if (ch < nc) q.pmx[ch] = scq;
        if (lane == 63) q.l2_post[g] = scq;
    }
    __syncthreads();
    for (int ch = tid; ch < nc; ch += PF_BS) {
        double runp = 0.0;
        for (int g = 0; g < ch / 64; ++g) runp = runp + q.l2_post[g];
        double offp = (ch % 64 == 0) ? 0.0 : q.pmx[ch - 1];
        A.rg_coffp[(size_t)slot * nc + ch] = runp + offp;
    }
    Ctrl::RowInfo& r = c->ri[slot];
    // generations that hold the ends of the row's count windows (both monotone along the sweep)
    // (the windows are a kernel argument: indexed with a uniform epoch so that they are read with scalar loads)
    // k_sweep / k_sweep_blc keep the windows in LDS: thread e takes epoch e (each of the two searches is a chain of dependent
    // loads; one epoch after the other, as the by-value form below has to, they were 20 us of a launch)
    constexpr bool W_IN_LDS = !std::is_same<KA, KArgs>::value;
    for (int e = W_IN_LDS ? (tid < E ? tid : E) : 0; e < E; e = W_IN_LDS ? E : e + 1) {
        const double wa = W.a[e], wb = W.b[e];
        if (tid != e) continue;
        int g = c->g_lo[e];
        while (g < G && A.gen_x0[(g + 1) % A.Gcap] <= wa) ++g;
        c->g_lo[e] = g;
        int h = c->g_hi[e];
        if (e >= W.first) {
            if (h < g) h = g;
            while (h < G && A.gen_x0[(h + 1) % A.Gcap] < wb) ++h;
            c->g_hi[e] = h;
        }
        r.g_lo[e] = g; r.g_hi[e] = h; r.wa[e] = wa; r.wb[e] = wb;
    }
    __syncthreads();
    if (tid == 0) {
        int gr = c->g_lo[0];
        for (int e = 1; e < E; ++e) gr = c->g_lo[e] < gr ? c->g_lo[e] : gr;
        c->g_retain = gr;
        // (the entry of two rows ago, or what the seed of the call left there; where the bookkeeping itself runs ahead of the counts with the
        // extend role -- run_sweep_split -- the oldest entry of the ring: the counts in flight are at most fifteen rows behind this one)
        c->g_safe = c->ri[(slot + PF_RING - (PL.nL == -3 ? PF_RING - 1 : 2)) & (PF_RING - 1)].g_retain;
        c->delayed_opp += W.b[E - 1] - W.a[E - 1];
        if (BIASED) {
            long long npend = 0;      // update_delayed_weight_count (count.cpp:395-397)
            for (int k = 0; k < nc; ++k) npend += A.rg_dpend[(size_t)slot * nc + k];
            c->delayed_count += (double)npend * (W.b[E - 1] - W.a[E - 1]);
        }
        c->first_epoch = W.first;
        c->count_active = W.first < E;
        if (W.first < E) c->pending_fin = 1;
        c->nbx_used = A.nbx;
        c->wq[PL.b_slot] = 0;                 // the next step deals out the ledger and count work of this row (pipe_roles)
        if (!(d.T > 0.0) && !c->err) c->err = ERR_ZERO_PROB;
        double logl = c->logl + dlog(d.T);
        c->logl = logl;
        const long long row = PL.b_row;
        A.tr_T[row] = d.T; A.tr_ess[row] = d.ess; A.tr_flag[row] = d.flag; A.tr_logl[row] = logl;
        c->cur_pos = PL.b_pos;
        c->T = d.T; c->inv_T = d.inv; c->S1 = d.S1; c->S2 = d.S2; c->ess = d.ess; c->u = d.u; c->flag = 0;
        r.T = d.T; r.inv_T = d.inv; r.S1 = d.S1; r.u = d.u; r.pos = PL.b_pos;
        r.n_res = n_res; r.gen = G; r.flag = d.flag; r.g_retain = gr; r.first = W.first; r.lver = c->lver;
        if (d.flag) {
            int ev = (int)n_res;
            if (ev < A.max_trace_events) A.ev_seg[ev] = (int)row;
            c->gen = G + 1;
            A.gen_x0[(G + 1) % A.Gcap] = PL.b_pos;
            c->n_resample = n_res + 1;
            c->lver ^= 1;                    // the ledger upkeep of this row (next launch) writes the other copy
            // an extend role that is a launch of its own is already writing offspring tables and log marks of up to PF_RING newer generations
            if (G + 1 - gr >= A.Gcap - 1 - (PL.ahead ? PF_RING : 0)) c->err = ERR_GEN_OVERFLOW;
            if (A.rec_trees && G + 2 >= A.Gcap) c->err = ERR_GEN_OVERFLOW;      // -arg keeps every generation
        }
        c->gen_prev = c->gen; c->nres_prev = c->n_resample;
        if (PL.b_set_cur >= 0) c->cur = PL.b_set_cur;
    }
}

// The draw table (one-population rows of k_sweep).  The random numbers of a slot are a function of (seed, slot, draw index)
// alone -- the stream belongs to the slot, not to the particle that sits in it -- so the numbers the slot's next genealogy
// updates will ask for can be made before they are asked for, by workgroups that are off the critical path: a third of
// an update's instructions are its two Philox blocks and the logarithms of two of the four uniforms.  Step s brings the
// table of every slot up to PF_DRAW_RING blocks past the counter the slot had at the start of row s (written by the extend
// role of step s - 1); the extend role of step s + 1 reads it.  What the extend role of step s reads at the same time lies
// below the old mark, what is written here at or above it, and the ring holds PF_DRAW_RING blocks: no entry is read and
// written in one launch.  A slot that outruns its table (more than sixteen updates in a row) computes its own numbers, and
// the table skips ahead.
template <class KA>
__device__ __forceinline__ void draw_role(const KA& A, int tb, int nT, int par) {
    const long long Np = A.Np;
    const unsigned long long* c_in = A.dt_ctr + (size_t)(par ^ 1) * Np;
    const unsigned long long* f_in = A.dt_filled + (size_t)(par ^ 1) * Np;
    unsigned long long* f_out = A.dt_filled + (size_t)par * Np;
    const unsigned long long seed = A.seed;
    for (long long p = (long long)tb * PF_BS + threadIdx.x; p < Np; p += (long long)nT * PF_BS) {
        const unsigned long long c0 = c_in[p], f_old = f_in[p];
        const unsigned long long to = c0 + PF_DRAW_RING;
        double2* tab = (double2*)A.dt_tab + (size_t)p * PF_DRAW_RING;
        for (unsigned long long cc = f_old > c0 ? f_old : c0; cc < to; ++cc) {
            double u0, u1;
            philox_pair(seed, (unsigned)p, 0u, cc, u0, u1);
            tab[(unsigned)cc & (PF_DRAW_RING - 1)] = make_double2(u0, -dlog(u1));
        }
        f_out[p] = to;
    }
}

// one block of the ledger upkeep of the row in ring slot PL.lc_slot (a resampling row): the new run list, or a share of the old ones
template <class KA>
__device__ __forceinline__ void pipe_ledger_item(const KA& A, const PipeLaunch& PL, const Ctrl::RowInfo& r, int lb) {
    const RunLists src = run_lists(A, r.lver), dst = run_lists(A, r.lver ^ 1);
    const int nblk = PL.nblk;              // particle blocks of 256 (the extend workgroups of the one-population kernels)
    if (lb < nblk) ledger_new_list(A, dst, A.rg_blkcnt + (size_t)PL.lc_slot * nblk * A.blk_gran, nblk, lb, r.gen, A.blk_gran);
    else ledger_update(A, lb - nblk, PL.nL - nblk, r.gen, r.g_retain, src, dst);
}

// count work item idx of that row: part cbx of cnb of the column of epoch e
template <int NM, bool EXACT, int P, class KA>
__device__ __forceinline__ void pipe_count_item(const KA& A, const PipeLaunch& PL, const Ctrl::RowInfo& r, int idx) {
    // the columns of the oldest epochs first: their lags are the shortest, their windows sit right behind the front where
    // nearly every particle is still its own ancestor, and their workgroups are the long ones -- dispatched last they were
    // what a launch ended with
    int e, cbx, cnb;
    if (A.cw_off && !(A.flags & 512)) {
        int j = 0, hi = A.E;                       // the column of workgroup idx: last j with cw_off[j] <= idx
        while (hi - j > 1) { const int mid = (j + hi) >> 1; if (A.cw_off[mid] <= idx) j = mid; else hi = mid; }
        e = A.E - 1 - j; cbx = idx - A.cw_off[j]; cnb = A.cw_off[j + 1] - A.cw_off[j];
    } else {
        e = (A.flags & 512) ? r.first + idx / PL.ncw : A.E - 1 - idx / PL.ncw;
        cbx = idx % PL.ncw; cnb = PL.ncw;
    }
    if (e >= A.E || e < r.first) return;
    if (A.flags & (1 << 22)) return;          // probe: the count workgroups do nothing at all
    const DState st = state_slot(A, PL.lc_slot);
    CountSrc Q;
    Q.w = st.w_post; Q.S = st.S; Q.xm = st.x_mark; Q.ml = st.mark_limit;
    Q.widx = A.rg_widx + (size_t)PL.lc_slot * A.Np;
    Q.widx_live = A.rg_widx + (size_t)PL.live_slot * A.Np;
    Q.scanp = A.rg_scanp + (size_t)PL.lc_slot * A.Np;
    Q.offp = A.rg_coffp + (size_t)PL.lc_slot * A.nc;
    Q.lists = run_lists(A, r.lver);
    Q.inv = r.inv_T; Q.G = r.gen; Q.g_lo = r.g_lo[e]; Q.g_hi = r.g_hi[e];
    count_body<NM, P, EXACT>(A, Q, e, r.wa[e], r.wb[e], cbx, cnb);
}

template <int NM, bool BIASED, bool EXACT, bool TREES, int P = 1, bool BLC = false, bool QUEUE = false, bool LEAN = false, class KA>
__device__ __forceinline__ void pipe_roles(const KA& A, long long s, const PipeLaunch& PL, const Windows& Wb) {
    const int bx = pf_bx();
    const int nb = PL.nb;
    // (LEAN: the launch of the ledger and count roles alone -- run_sweep_split -- is compiled without the other three: their registers and LDS
    // are what would keep it from four workgroups per compute unit)
    if constexpr (!LEAN) {
        if (bx < nb) {
            if constexpr (P == 1) { if (PL.row.extend || PL.row.complete) extend_reg_body<NM, BIASED, EXACT, TREES, true>(A, s, 0, PL.row); }
            return;
        }
        if (bx == nb) {
            if (PL.b_slot < 0 || PL.nL == -2) return;          // nL = -2: the extend launch of a split step (-3: one that keeps the bookkeeping)
            extern __shared__ double smem[];
            PipeLds q = pipe_carve(smem + (2 * PF_EPAD + A.E + 2 * PF_BIAS_MAX + 3), A.nc);
            pipe_bookkeeping<BIASED>(A, q, PL, Wb);
            return;
        }
        if (bx - (nb + 1) < PL.nT) {
            if constexpr (P == 1) draw_role(A, bx - (nb + 1), PL.nT, PL.row.draws - 1);
            return;
        }
    } else {
        if (bx <= nb) return;
    }
    if (PL.lc_slot < 0 || PL.nL < -1) return;
    const Ctrl* c = A.ctrl;
    const Ctrl::RowInfo& r = c->ri[PL.lc_slot];
    const int lb = bx - (nb + 1) - PL.nT;
    // The ledger and count work of the step: items -- a ledger block of a resampling row, or part cbx of the count column of an
    // epoch.  Without pf_params.count_workers every item is a workgroup of this launch (ledger blocks first, then the columns,
    // oldest epoch first).
    if constexpr (!QUEUE) {
        if (lb < PL.nL) { if (r.flag) pipe_ledger_item(A, PL, r, lb); }
        else pipe_count_item<NM, EXACT, P>(A, PL, r, lb - PL.nL);
    } else {
        // With it, `workers` workgroups take the items -- the columns first, then the ledger's blocks -- off a counter until none is
        // left: an item writes the same accumulators whoever runs it, the sums do not change.  What that saves several chunks per
        // GPU: the 232 ledger workgroups per chunk and step that find no resampling and leave (2 us of a slot each), the prologue
        // of 260 count workgroups, and the dispatcher's strict order -- workgroup i + 1 is not placed before workgroup i, so one
        // full XCD holds up the other seven while long count workgroups run (profiles/round4/wg_trace.md).  A kernel instance of
        // its own (k_sweep4q): in the static form the count body is the last thing a workgroup does and nothing is live across it;
        // what this loop needs after an item comes back from LDS.
        __shared__ int s_wk[8];
        if (lb >= PL.workers) return;
        if (threadIdx.x == 0) { s_wk[0] = PL.lc_slot; s_wk[1] = PL.live_slot; s_wk[2] = PL.nL; s_wk[3] = PL.nblk; s_wk[4] = PL.ncw; }
        const KA* Ap = &A;
        for (;;) {
            // (the argument block behind a pointer the compiler cannot see through: otherwise it loads every field the item bodies use once,
            // before the loop, keeps them all in registers across the items and spills vector registers to scratch memory to do so)
            asm volatile("" : "+s"(Ap));
            const KA& A2 = *Ap;
            __syncthreads();
            if (threadIdx.x == 0) s_wk[6] = (int)__hip_atomic_fetch_add(&A2.ctrl->wq[s_wk[0]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            PipeLaunch Q2;
            Q2.lc_slot = s_wk[0]; Q2.live_slot = s_wk[1]; Q2.nL = s_wk[2]; Q2.nblk = s_wk[3]; Q2.ncw = s_wk[4];
            const int it = s_wk[6];
            const Ctrl::RowInfo& r2 = A2.ctrl->ri[Q2.lc_slot];
            const int ncnt = r2.first < A2.E ? (A2.cw_off ? A2.cw_off[A2.E - r2.first] : (A2.E - r2.first) * Q2.ncw) : 0;
            if (it >= ncnt + (r2.flag ? Q2.nL : 0)) return;                  // (every workgroup gets here: the counter only grows)
            // (inlined: as real calls the item bodies save and restore 124 registers per item in scratch memory -- 5.7e4 segments/s
            // against 8.0e4 for eight chunks with 264 workers each, and 9.4e4 without the queue)
            if (it < ncnt) pipe_count_item<NM, EXACT, P>(A2, Q2, r2, it);
            else pipe_ledger_item(A2, Q2, r2, it - ncnt);
        }
    }
}

template <int NM, bool BIASED, bool EXACT, bool TREES>
__global__ __launch_bounds__(PF_BS) void k_pipe(KArgs A, long long s, PipeLaunch PL, Windows Wb) {
    pipe_roles<NM, BIASED, EXACT, TREES>(A, s, PL, Wb);
}

// ------------------------------------------------------------------ k_sweep: several chunks per launch
// The same four roles as k_pipe, for C independent chunks at once: blockIdx.x is the chunk (pf_bx(), pf_device.h), each with its own argument
// block in device memory (SweepChunk::A, read through the constant address space: every field is a scalar load where it
// is used, nothing of the block is a kernel argument any more), its own Ctrl, rings and event log.  All chunks step in
// lockstep: launch t handles row s = s_begin + t of every chunk that still has one (then its two flush steps).  What
// k_pipe is told per launch by the host is derived here: the plan of the step (sweep_plan = the `launch` lambda of
// run_pipeline) from (s, s_begin, s_last), and the count windows of row s - 1 by the bookkeeping workgroup from the
// chunk's own window state (sweep_windows = host_windows, operation for operation), so the host only supplies t.
// One chunk per GPU uses a sixth of the SIMDs; the reference's data parallelism is one process per chunk, all at once
// (smcsmc/model.py:1094-1098), which is what a grid over chunks is here.
// extract_and_update_count's window rule (count.cpp:363-385) for the row that ends at `pos`: host_windows() on the device,
// the same operations in the same order on the same doubles; the window state lives in Ctrl::counted_to.  Called by all
// threads of the bookkeeping workgroup; W is in LDS.
template <class KA>
__device__ __forceinline__ void sweep_windows(const KA& A, Ctrl* c, double pos, Windows& W) {
    const int e = threadIdx.x, E = A.E;
    if (e < 64) {
        const bool in = e < E;
        const double lagging = in ? A.lags[e] : 0.0;
        const double x_end = pos - lagging;
        const double ct = in ? c->counted_to[e] : 0.0;
        const bool small = (x_end - ct) < lagging * 0.1;
        const unsigned long long moved = __ballot(in && !small);
        const int first = moved ? (int)__builtin_ctzll(moved) : E;
        if (in) {
            const double b = (small && first > e) ? ct : x_end;
            W.a[e] = ct; W.b[e] = b;
            c->counted_to[e] = b;
        }
        if (e == 0) { W.first = first; W.end_data = 0; }
    }
    __syncthreads();
}

// where a workgroup of a traced step (pf_set_wg_trace) leaves its four words: start and end on the 100 MHz clock, HW_ID | XCC_ID << 32,
// index within the chunk | chunk << 32
__device__ __forceinline__ unsigned long long* wg_trace_slot(SweepChunkC& ch, long long t) {
    if (!ch.trace || t < ch.trace_t0 || t >= ch.trace_t0 + ch.trace_n) return nullptr;
    const size_t lin = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
    return ch.trace + ((size_t)(t - ch.trace_t0) * (size_t)ch.trace_stride + lin) * 4;
}

template <int NM, bool BIASED, bool EXACT, bool TREES, bool HANDOFF = false, bool TRACE = false, bool QUEUE = false>
__device__ __forceinline__ void sweep_kernel_body(const SweepChunk* tab_g, long long t, int nb) {
    SweepChunkC* tab = (SweepChunkC*)tab_g;
    SweepChunkC& ch = tab[pf_chunk()];
    KArgsC& A = ch.A;
    const long long s = ch.s_begin + t;
    __shared__ Windows W;           // written and read by the bookkeeping workgroup only
    bool ok = true;
    if constexpr (TRACE) {
        if (threadIdx.x == 0) {
            if (unsigned long long* w = wg_trace_slot(ch, t)) {
                w[0] = wall_clock64();
                w[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
                w[3] = (unsigned long long)(unsigned)pf_bx() | ((unsigned long long)(unsigned)pf_chunk() << 32);
            }
        }
    }
    if constexpr (HANDOFF) {
        // this launch was enqueued without waiting for the one of step t - 1 to end: its workgroups wait for that launch's workgroups
        // to have arrived (state, scans, partials and draw table of row s - 1 are then visible), and -- ring reuse -- for the
        // bookkeeping / ledger / count launch of step t - 14 to have ended
        Ctrl* c = A.ctrl;
        if (t >= PF_RING - 2) ok = sweep_wait_ge(c, (const unsigned*)&c->blc_step, (unsigned)(t - (PF_RING - 3)));      // (nothing is read from that launch)
        if (ok && t > 0) { ok = sweep_wait_ge(c, &c->xt_done[(t - 1) & (PF_RING - 1)], (unsigned)((t - 1) / PF_RING + 1) * (unsigned)ch.xt_wgs); sweep_acquire(); }
    }
    PipeLaunch PL;
    if (ok && sweep_plan(ch, s, nb, PL)) {
        if (ch.split == 1) PL.nL = -2;                     // extend and draw roles only: the other roles are k_sweep_blc's
        else {
            if (ch.split == 2) PL.nL = -3;                 // ... the bookkeeping too: only ledger and counts are k_sweep_blc's (run_sweep_split)
            if (pf_bx() == nb && PL.b_slot >= 0) sweep_windows(A, A.ctrl, PL.b_pos, W);
        }
        pipe_roles<NM, BIASED, EXACT, TREES, 1, false, QUEUE>(A, s, PL, W);
    }
    if constexpr (HANDOFF) sweep_arrive(&A.ctrl->xt_done[t & (PF_RING - 1)]);
    if constexpr (TRACE) {
        __syncthreads();
        if (threadIdx.x == 0) if (unsigned long long* w = wg_trace_slot(ch, t)) w[1] = wall_clock64();
    }
}
template <int NM, bool BIASED, bool EXACT, bool TREES>
__global__ __launch_bounds__(PF_BS) void k_sweep(const SweepChunk* tab_g, long long t, int nb) {
    sweep_kernel_body<NM, BIASED, EXACT, TREES>(tab_g, t, nb);
}
// The headline instance (at most four haplotypes, no focused sampling) with the occupancy stated: three workgroups per compute
// unit, i.e. at most 168 registers a thread -- several chunks per GPU lose 15 % with two.  Left to itself
// the register allocator aims at the occupancy the kernel's LDS allows: that gave 168 while the count roles' LDS was 8 KB
// larger and 172-174 since round 4.  (The instances for more haplotypes or focused sampling need about 190 registers and would
// spill to scratch memory under the same attribute; they stay as they are.)
template <bool EXACT, bool TREES>
__global__ __launch_bounds__(PF_BS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_sweep4(const SweepChunk* tab_g, long long t, int nb) {
    sweep_kernel_body<4, false, EXACT, TREES>(tab_g, t, nb);
}
// the extend / draw launch of run_sweep_flags: the same body behind a wait for the previous step's arrivals (a kernel of its own:
// the few values the wait keeps alive would tip k_sweep4 into scratch memory)
template <bool EXACT>
__global__ __launch_bounds__(PF_BS) void k_sweep4h(const SweepChunk* tab_g, long long t, int nb) {
    sweep_kernel_body<4, false, EXACT, false, true>(tab_g, t, nb);
}

// the headline instance with the ledger and count work of a step taken off a queue (pf_params.count_workers; the traced form too)
template <bool EXACT, bool TRACE>
__global__ __launch_bounds__(PF_BS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_sweep4q(const SweepChunk* tab_g, long long t, int nb) {
    sweep_kernel_body<4, false, EXACT, false, false, TRACE, true>(tab_g, t, nb);
}
// the headline instance with a time stamp at either end of every workgroup (pf_set_wg_trace: where do the slots of a step go?)
template <bool EXACT>
__global__ __launch_bounds__(PF_BS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_sweep4t(const SweepChunk* tab_g, long long t, int nb) {
    sweep_kernel_body<4, false, EXACT, false, false, true>(tab_g, t, nb);
}

// The bookkeeping, ledger and count roles of a step as a launch of their own: what the structured models run on the counting
// stream beside their extend launch (k_sweep_xmp, pf_mp.hip).  The extend workgroups of those models carry their trees'
// migration events in LDS (61 KB per 64 particles at the default capacity); in one launch every count workgroup would be
// given the same allocation and one would fit a CU.
template <int NM, int P, bool BIASED, bool EXACT = false, bool LEAN = false, bool QUEUE = false>
__device__ __forceinline__ void sweep_blc_body(const SweepChunk* tab_g, long long t) {
    SweepChunkC* tab = (SweepChunkC*)tab_g;
    SweepChunkC& ch = tab[pf_chunk()];
    KArgsC& A = ch.A;
    const long long s = ch.s_begin + t;
    __shared__ Windows W;
    Ctrl* c = A.ctrl;
    PipeLaunch PL;
    if (sweep_plan(ch, s, 0, PL)) {
        PL.nT = 0;                                         // the draw role rides with the extend launch
        if (ch.split == 2) PL.b_slot = -1;                 // ... and so does the bookkeeping (run_sweep_split)
        if (pf_bx() == 0 && PL.b_slot >= 0) sweep_windows(A, c, PL.b_pos, W);
        pipe_roles<NM, BIASED, EXACT, false, P, true, QUEUE, LEAN>(A, s, PL, W);
    }
    if (ch.handoff) {
        // run_sweep_flags: the extend launch of step t + 14 overwrites ring slots this launch read; it polls Ctrl::blc_step, which the
        // last workgroup of this launch to finish advances (launches of this kind run one after the other; nothing they WRITE is read
        // by an extend launch, so no release is needed -- a release per workgroup would write the L2 back a thousand times a step)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned before = __hip_atomic_fetch_add(&c->blc_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1 == gridDim.x) {
                __hip_atomic_store(&c->blc_arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&c->blc_step, (int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int NM, int P, bool BIASED>
__global__ __launch_bounds__(PF_BS) void k_sweep_blc(const SweepChunk* tab_g, long long t) {
    sweep_blc_body<NM, P, BIASED>(tab_g, t);
}
// the ledger and count roles of several chunks beside their extend launches (run_sweep_split): at most four haplotypes, no focused
// sampling, and no dynamic LDS -- four workgroups to a compute unit (128 registers, 36.6 KB)
template <bool EXACT>
__global__ __launch_bounds__(PF_BS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sweep_blc4(const SweepChunk* tab_g, long long t) {
    sweep_blc_body<4, 1, false, EXACT, true>(tab_g, t);
}
// ... with the ledger and count items of a step taken off a queue by a fixed number of workgroups (pf_params.count_workers)
template <bool EXACT>
__global__ __launch_bounds__(PF_BS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sweep_blc4q(const SweepChunk* tab_g, long long t) {
    sweep_blc_body<4, 1, false, EXACT, true, true>(tab_g, t);
}

// first step of a call: the seed of k_pipe_seed, and the chunk's window state
__global__ void k_sweep_seed(const SweepChunk* tab_g) {
    SweepChunkC* tab = (SweepChunkC*)tab_g;
    SweepChunkC& ch = tab[blockIdx.x];
    KArgsC& A = ch.A;
    if (ch.s_last < ch.s_begin) return;              // nothing to do for this chunk in this call
    Ctrl* c = A.ctrl;
    const int slot = (int)((ch.s_begin + PF_RING - 1) & (PF_RING - 1));
    if (threadIdx.x == 0) {
        Ctrl::RowInfo& r = c->ri[slot];
        r.n_res = c->n_resample; r.gen = c->gen; r.flag = 0; r.lver = c->lver; r.g_retain = c->g_retain; r.first = A.E;
        r.inv_T = 1.0; r.T = 1.0; r.S1 = 0.0; r.u = 0.0; r.pos = c->cur_pos;
        c->xr[slot].n_res = c->n_resample; c->xr[slot].gen = c->gen; c->xr[slot].flag = 0;
    }
    if ((int)threadIdx.x < A.E) c->counted_to[threadIdx.x] = ch.counted_to[threadIdx.x];
    if ((int)threadIdx.x < PF_RING) { c->xt_done[threadIdx.x] = 0; c->wq[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { c->blc_arrive = 0; c->blc_step = 0; }
    if (ch.nT > 0) {
        // the draw table starts every call empty, from the counters the slots have now
        const size_t par = (size_t)((ch.s_begin + 1) & 1);          // parity of row s_begin - 1
        for (long long p = threadIdx.x; p < A.Np; p += blockDim.x) {
            A.dt_ctr[par * A.Np + p] = A.rng_ctr[p];
            A.dt_filled[par * A.Np + p] = 0;
        }
    }
}

__global__ __launch_bounds__(PF_BS) void k_ledger(KArgs A, int nblocks) {
    ledger_body(A, A.sp, nblocks, (int)blockIdx.x, (int)gridDim.x);
}

__global__ __launch_bounds__(PF_BS) void k_decide(KArgs A, long long s, int mode, Windows W, int nblocks) {
    decide_body(A, s, mode, W, nblocks);
}

// k_decide of row s with the ledger maintenance of row s-1 riding along in extra workgroups (single-stream pipeline:
// it has to follow the counts of row s-1, which ran inside k_row(s), and precede those of row s)
// Launched with PF_LEDGER_MAXT threads per workgroup: the re-basing of a run list is bound by memory round trips per
// tile, and with 1024 threads the longest list is one tile.  The decide, bookkeeping and run-list workgroups are
// written for PF_BS threads; their other wavefronts leave at once.
// Hardware-specific: the wavefronts that leave do so before the first barrier of the body, and on gfx9-class hardware
// (gfx950 included) a wavefront that has ended no longer counts towards s_barrier, so the remaining four synchronise
// among themselves; HIP itself leaves a barrier under divergent exit undefined.  Since round 2 this kernel is only
// launched with PF_DEBUG_TWO_LAUNCH (the A/B form of the row pipeline); tests/test_gpu_headline.py
// ::test_rings_wrap_many_times[...-8] pins the behaviour on the target.
__global__ __launch_bounds__(PF_LEDGER_MAXT) void k_decide_ledger(KArgs A, long long s, int mode, Windows W, int nblocks, int ledger_nbt) {
    if ((int)blockIdx.x <= nblocks) {
        if (threadIdx.x < PF_BS) decide_body(A, s, mode, W, nblocks);
    } else {
        const int bx = (int)blockIdx.x - (nblocks + 1);
        if (bx >= nblocks || threadIdx.x < PF_BS) ledger_body(A, A.sp ^ 1, nblocks, bx, ledger_nbt);
    }
}

__global__ __launch_bounds__(PF_BS) void k_resample(KArgs A, long long s, int nblocks) {
    Ctrl* c = A.ctrl;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c->gen_prev = c->gen; c->nres_prev = c->n_resample; }
    const long long Np = A.Np;
    const long long i = (long long)blockIdx.x * PF_BS + threadIdx.x;
    const double inv = c->inv_T;
    if (!c->flag) {
        if (i >= Np) return;
        const DState st = state_slot(A, __builtin_amdgcn_readfirstlane(c->cur));
        st.w_post[i] *= inv;
        st.w_pilot[i] *= inv;
        return;
    }
    if (i >= Np) return;
    const int n = A.n;
    const int G = c->gen - 1;                  // the generation that ends here (k_decide already advanced gen)
    const DState src = state_slot(A, __builtin_amdgcn_readfirstlane(c->cur ^ 1));
    const DState dst = state_slot(A, __builtin_amdgcn_readfirstlane(c->cur));
    const int* lo = A.lo + (size_t)(G % A.Gcap) * (Np + 1);
    const double pos = c->cur_pos;
    // ---- role 1: old slot i closes its stretch if it has offspring ----
    unsigned widx = A.widx[i];
    if (lo[i + 1] > lo[i]) {
        double* rec = rec_ptr(A, i, widx);
        rec[0] = src.x_mark[i];
        rec[1] = pos;
        rec[2] = 0.0; rec[3] = 0.0;
        rec[4] = __longlong_as_double((long long)make_meta(1, src.mark_limit[i], -1, n));
        for (int r = 0; r < n - 1; ++r) rec[5 + r] = src.S[(size_t)r * Np + i];
        ++widx;
        A.widx[i] = widx;
    }
    A.gstart[(size_t)((G + 1) % A.Gcap) * Np + i] = widx;
    // ---- role 2: new slot q = i copies its parent (table written by k_decide) ----
    const int q = (int)i;
    const long long a = A.parent[q];
    int ev = (int)c->n_resample - 1;
    if (ev < A.max_trace_events) A.ev_parents[(size_t)ev * Np + q] = (int)a;
    for (int r = 0; r < n - 1; ++r) {
        dst.S[(size_t)r * Np + q] = src.S[(size_t)r * Np + a];
        dst.C[(size_t)(2 * r) * Np + q] = src.C[(size_t)(2 * r) * Np + a];
        dst.C[(size_t)(2 * r + 1) * Np + q] = src.C[(size_t)(2 * r + 1) * Np + a];
    }
    if (A.P > 1) {
        for (int r = 0; r < n - 1; ++r) dst.Pn[(size_t)r * Np + q] = src.Pn[(size_t)r * Np + a];
        const int nmv = src.nm[a];
        dst.nm[q] = nmv;
        for (int k = 0; k < nmv; ++k) {
            dst.Mt[(size_t)k * Np + q] = src.Mt[(size_t)k * Np + a];
            dst.Mb[(size_t)k * Np + q] = src.Mb[(size_t)k * Np + a];
            dst.Mq[(size_t)k * Np + q] = src.Mq[(size_t)k * Np + a];
        }
    }
    // weights: normalise, then adjustment = sum / (N * pilot)   (pc.cpp:350-351)
    double wp = src.w_post[a] * inv;
    double wq = src.w_pilot[a] * inv;
    double sumn = c->S1 * inv;
    double adj = sumn / ((double)Np * wq);
    dst.w_post[q] = wp * adj;
    dst.w_pilot[q] = wq * adj;
    double Lt = src.Ltree[a];
    dst.Ltree[q] = Lt;
    dst.x_mark[q] = pos;
    dst.mark_limit[q] = src.mark_limit[a];
    if (A.apf > 0) dst.lookahead[q] = src.lookahead[a];
    if (A.g_K > 0) dst.ridx[q] = src.ridx[a];
    if (A.n_bias > 0 || A.g_K > 0) {
        // the copy constructor copies the pending factors (particle.cpp:122-123)
        int dc = src.dcount[a];
        dst.dcount[q] = dc;
        dst.total_delayed[q] = src.total_delayed[a];
        for (int k = 0; k < dc; ++k) {
            dst.dpos[(size_t)k * Np + q] = src.dpos[(size_t)k * Np + a];
            dst.dfac[(size_t)k * Np + q] = src.dfac[(size_t)k * Np + a];
            dst.ddelta[(size_t)k * Np + q] = src.ddelta[(size_t)k * Np + a];
            dst.dk[(size_t)k * Np + q] = src.dk[(size_t)k * Np + a];
        }
    }
    double nb = src.next_base[a];
    if (q != lo[a] && pos < A.L) {
        // a copy: draw a fresh recombination position from slot q's own stream (pc.cpp:357-368)
        Lane ln;
        ln.E = A.E; ln.n = n; ln.L = A.L; ln.mu = A.mu; ln.rho = A.rho; ln.seed = A.seed;
        ln.slot = (unsigned)q; ln.stream = 0; ln.ctr = A.rng_ctr[q]; ln.ebuf = A.ebuf[q]; ln.Ltree = Lt;
        ln.S = nullptr; ln.C = nullptr; ln.T = nullptr; ln.I = nullptr; ln.RF = nullptr;
        if (A.g_K > 0) {
            // with a guide: the rate of the copy's segment, the draw limited to the segment
            const int ri = src.ridx[a];
            ln.rho = A.g_rho[ri];
            if (ri + 1 < A.g_K && A.g_pos[ri + 1] < A.L) ln.L = A.g_pos[ri + 1];
        }
        nb = sample_next_base(ln, pos);
        A.rng_ctr[q] = ln.ctr;
        A.ebuf[q] = ln.ebuf;
    }
    dst.next_base[q] = nb;
}

// recompute the level-1 partials from the stored weights (used by pf_finish: smcsmc.cpp:371)
__global__ __launch_bounds__(PF_BS) void k_partials(KArgs A) {
    const Ctrl* c = A.ctrl;
    const long long p = (long long)blockIdx.x * PF_BS + threadIdx.x;
    const bool active = p < A.Np;
    const int lane = threadIdx.x & 63;
    const DState st = state_slot(A, __builtin_amdgcn_readfirstlane(c->cur));
    double w_post = active ? st.w_post[p] : 0.0;
    double w_pilot = active ? st.w_pilot[p] : 0.0;
    if (active) {
        for (int r = 0; r < A.n - 1; ++r) A.snap_S[A.sp][(size_t)r * A.Np + p] = st.S[(size_t)r * A.Np + p];
        A.snap_w[A.sp][p] = w_post; A.snap_xm[A.sp][p] = st.x_mark[p]; A.snap_ml[A.sp][p] = st.mark_limit[p];
        A.snap_widx[A.sp][p] = A.widx[p];
    }
    double sp = wave_tree_sum(w_post);
    double sq = wave_tree_sum(w_pilot * w_pilot);
    double sc = wave_hs_scan(w_pilot, lane);
    double scp = wave_hs_scan(w_post, lane);
    double scm = wave_max_scan_d(sc, lane);     // running max of the pilot scan (a parallel FP scan need not be monotone)
    long long chunk = p >> 6;
    if (active) { A.scan1[p] = sc; A.scanp2[A.sp][p] = scp; A.scan1m[p] = scm; }
    if (lane == 63 && chunk < (A.Np + 63) / 64) {
        A.chunk_post[chunk] = sp;
        A.chunk_sq[chunk] = sq;
        A.chunk_pil[chunk] = sc;
        A.chunk_pp[chunk] = scp;
        A.chunk_mx1[chunk] = scm;
    }
}

// ------------------------------------------------------------------ k_calibrate
// calculate_median_survival_distances (smcsmc.cpp:169-263): one prior ARG per lane; evolve it along
// the sequence without data until every internal node of the initial tree has been removed (or
// 0.6 L is reached) and report, per original node, its epoch and the position where it disappeared.
// NM = 0: the recombination loop runs on the LDS tree (any nsam); NM = 4 / 8: on the register tree of the extend
// kernels (nsam <= NM), same arithmetic operation for operation, about half the instructions.  The epoch tables of
// the register path sit behind the LDS-tree block.
template <int NM>
__global__ __launch_bounds__(PF_BS) void k_calibrate(KArgs A, unsigned long long seed, long long rep0, long long nrep,
                                                     int* out_epoch, double* out_dist) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    load_model(A, m);
    double* sT = (double*)((char*)smem + smem_bytes(A.n, A.E));
    double* sH = sT + PF_EPAD;
    double* sI = sH + PF_EPAD;
    if (NM > 0)
        for (int e = threadIdx.x; e < PF_EPAD; e += blockDim.x) {
            sT[e] = e < A.E ? A.T[e] : PF_INF;
            sH[e] = e < A.E ? A.Hc[e] : PF_INF;
            if (e < A.E) sI[e] = A.inv2N[e];
        }
    __syncthreads();
    long long r = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (r >= nrep) return;
    const int n = A.n;
    Lane ln = make_lane(A, m, rep0 + r);
    ln.seed = seed;
    ln.stream = 2;
    ln.ebuf = -dlog(uni(ln));
    int root = 0;
    for (int i = 1; i < n; ++i) {
        int ni = i - 1;
        double tc = coalesce_up(ln, [&](int k) { return LS(ln, k); }, ni, i, 0.0);
        int pr = -1, ps = 0;
        int k = lineages_at(ln, ni, tc, -1, &pr, &ps);
        bool above_root = (ni == 0) || (tc >= LS(ln, ni - 1));
        int kk = above_root ? 1 : k;
        double u = uni(ln);
        int idx = min((int)(u * (double)kk), kk - 1);
        if (above_root) insert_node(ln, ni, tc, i, -1, 0, root);
        else { lineages_at(ln, ni, tc, idx, &pr, &ps); insert_node(ln, ni, tc, i, pr, ps, root); }
        root = n + ni;
    }
    ln.Ltree = tree_length(ln, n);
    // original internal-node heights live in the t0 scratch column of this lane
    double* orig = m.t0 + threadIdx.x;
    int alive = n - 1;
    for (int j = 0; j < n - 1; ++j) {
        orig[j * PF_BS] = LS(ln, j);
        out_epoch[r * (n - 1) + j] = epoch_of(ln, LS(ln, j));
        out_dist[r * (n - 1) + j] = -1.0;
    }
    unsigned alive_mask = (1u << (n - 1)) - 1u;
    double next = sample_next_base(ln, 0.0);
    const double stop = A.L * 0.6;
    if constexpr (NM > 0) {
        RTree<NM> t;
#pragma unroll
        for (int rr = 0; rr < RTree<NM>::NI; ++rr) {
            t.S[rr] = 0.0; t.C0[rr] = 0; t.C1[rr] = 0;
            if (rr < n - 1) { t.S[rr] = LS(ln, rr); t.C0[rr] = LC(ln, rr, 0); t.C1[rr] = LC(ln, rr, 1); }
        }
        RCtx cx;
        cx.T = sT; cx.I = sI; cx.H = sH; cx.E = A.E; cx.n = n; cx.L = A.L; cx.mu = A.mu; cx.rho = A.rho;
        cx.seed = seed; cx.slot = ln.slot; cx.stream = 2; cx.ctr = ln.ctr; cx.ebuf = ln.ebuf; cx.Ltree = ln.Ltree;
        cx.nb = 1; cx.bH = nullptr; cx.bS = nullptr; cx.last_iw = 1.0; cx.want_desc = false; cx.want_desc_new = false; cx.last_desc = 0; cx.last_desc_new = 0;
        cx.vbc = nullptr; cx.upd_fac = 1.0;
        cx.gK = 0; cx.gpos = nullptr; cx.grho = nullptr; cx.gleaf = nullptr; cx.last_rbiw = 1.0; cx.ridx = 0; cx.g_rp = 0; cx.g_sb = 0;
        while (alive > 0 && next < stop) {
            const double x = next;
            double h, tc, sp;
            bool changed;
            r_genealogy_update<NM, false>(cx, t, &h, &tc, &sp, &changed);
            if (changed) {
                for (int j = 0; j < n - 1; ++j)
                    if (((alive_mask >> j) & 1u) && orig[j * PF_BS] == sp) {
                        out_dist[r * (n - 1) + j] = x;
                        alive_mask &= ~(1u << j);
                        --alive;
                        break;
                    }
            }
            next = r_sample_next_base<true>(cx, x);
        }
        return;
    }
    while (alive > 0 && next < stop) {
        double x = next;
        double h, tc, sp;
        bool changed;
        genealogy_update(ln, &h, &tc, &sp, &changed);
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
        ln.uqn = 0;
    }
}

// ------------------------------------------------------------------ k_lookahead
// update_lookahead_likelihood (pc.cpp:227-240) with ForestState::includeLookaheadLikelihood (particle.cpp:439-617):
// the previous look-ahead factor leaves the pilot weight, the new one enters it.  Runs after k_extend (the pilot
// weight is a product of the same factors in either order) and rewrites the pilot part of the per-wavefront
// partials that k_decide consumes.  One population.
static size_t smem_bytes_la(int n, int E) { return smem_bytes(n, E) + (size_t)PF_BS * (2 * n * 8 + n * 4); }

__global__ __launch_bounds__(PF_BS) void k_lookahead(KArgs A, long long row) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    double* s_lh = (double*)((char*)smem + smem_bytes(A.n, A.E)) + threadIdx.x;       // [n] stride PF_BS
    double* s_mp = s_lh + (size_t)A.n * PF_BS;                                           // [n]
    int* s_par = (int*)((double*)((char*)smem + smem_bytes(A.n, A.E)) + (size_t)2 * A.n * PF_BS) + threadIdx.x;   // [n]
    load_model(A, m);
    __syncthreads();
    const Ctrl* c = A.ctrl;
    const int n = A.n, Q = A.la_Q, D = A.la_D;
    const long long p = (long long)blockIdx.x * PF_BS + threadIdx.x;
    const bool active = p < A.Np;
    const int lane = threadIdx.x & 63;
    double w_pilot = 0.0;
    if (active) {
        const DState st = state_slot(A, __builtin_amdgcn_readfirstlane(c->cur));
        Lane ln = make_lane(A, m, p);
        for (int r = 0; r < n - 1; ++r) {
            LS(ln, r) = st.S[(size_t)r * A.Np + p];
            LC(ln, r, 0) = st.C[(size_t)(2 * r) * A.Np + p];
            LC(ln, r, 1) = st.C[(size_t)(2 * r + 1) * A.Np + p];
        }
        const double Ltree = st.Ltree[p];
        const double* fsd = A.la_fsd + (size_t)row * n;
        const double* rmr = A.la_rmr + (size_t)row * n;
        const int8_t* unph = A.la_unph + (size_t)row * n;
        const double recomb_rate = A.rho, mut_rate = A.mu;
        double likelihood = 1.0;
        const double rho_tbl = 2 * recomb_rate * (n - 1) / n;
        for (int i = 0; i < n; ++i) {
            int pr = -1;
            for (int r = 0; r < n - 1 && pr < 0; ++r)
                if (LC(ln, r, 0) == i || LC(ln, r, 1) == i) pr = r;
            double hgt = LS(ln, pr);
            if (A.P > 1) {
                // structured models: the first local node above a leaf may be a migration on its branch (the events
                // of a particle are sorted by time, so the first hit is the lowest)
                const int nmig = st.nm[p];
                for (int mi = 0; mi < nmig; ++mi)
                    if (st.Mb[(size_t)mi * A.Np + p] == i) { hgt = st.Mt[(size_t)mi * A.Np + p]; pr = -2 - mi; break; }
            }
            s_par[i * PF_BS] = pr;
            s_lh[i * PF_BS] = hgt;
            s_mp[i * PF_BS] = 0.0;
        }
        for (int i = 0; i < n; i++) {
            double pp = 0;
            const double si = fsd[i];
            double li = s_lh[i * PF_BS];
            const bool up = unph[i] != 0;
            if (up) li += s_lh[(i + 1) * PF_BS];
            const double rel_mut_rate = rmr[i];
            const double li_mu = li * mut_rate * rel_mut_rate;
            s_mp[i * PF_BS] = li_mu;
            if (up) s_mp[(i + 1) * PF_BS] = li_mu;
            for (int r = 0; r < 2; r++) {
                const double rel = r == 0 ? 1.0 : 0.5;
                const double li_rho = li * rho_tbl * rel;
                const double fe = fastexp_approx(-(li_rho + li_mu) * fabs(si));
                for (int q = 0; q < Q; ++q) {
                    double qbot = (q == 0 ? 0.0 : A.la_q[q - 1]);
                    double qtop = (q == Q - 1 ? 1.0 : A.la_q[q]);
                    double l_prime = A.la_tbl[i * Q + q];
                    double lprime_mu = l_prime * mut_rate * rel_mut_rate;
                    double div = (li_rho + li_mu - lprime_mu);
                    if (fabs(div) < (li_rho + li_mu + lprime_mu) * 1e-5) lprime_mu = lprime_mu * 1.0001;
                    if (si > 0) {
                        pp += 0.5 * (qtop - qbot) * ((li_rho * lprime_mu * fastexp_approx(-lprime_mu * si) +
                                                      (li_mu - lprime_mu) * (li_rho + li_mu) * fe)
                                                     / (li_rho + li_mu - lprime_mu));
                    } else {
                        pp += 0.5 * (qtop - qbot) * ((li_rho * fastexp_approx(-lprime_mu * (-si)) +
                                                      (li_mu - lprime_mu) * fe)
                                                     / (li_rho + li_mu - lprime_mu));
                    }
                }
            }
            likelihood *= pp;
            if (up) i++;
        }
        if (A.apf >= 2) {
            double l_mean = 0.0;
            for (int i = 0; i < n; i++) l_mean += A.la_tbl[i * Q + (Q - 1)] / n;
            const double rho_c = 4 * recomb_rate * (n - 2) / n;
            const double rhoprime_c = recomb_rate * (n - 1);
            const double p_equilibrium = 2.0 / (3 * (n - 1));
            const int nd = A.la_nd[row];
            for (int k = 0; k < nd; ++k) {
                const int8_t* di = A.la_didx + ((size_t)row * D + k) * 4;
                const double fed = A.la_ddist[((size_t)row * D + k) * 2], led = A.la_ddist[((size_t)row * D + k) * 2 + 1];
                int ph1, ph2;
                for (ph1 = 0; ph1 <= di[2]; ph1++) {
                    for (ph2 = 0; ph2 <= di[3]; ph2++) {
                        int a = di[0] + ph1, b = di[1] + ph2;
                        if (s_par[a * PF_BS] == s_par[b * PF_BS]) {
                            double l = s_lh[a * PF_BS];
                            double pp = 0;
                            for (int r = 0; r < 2; r++) {
                                double exp_rho = fastexp_approx(-rho_c * (r == 0 ? 1.0 : 0.5) * l * led);
                                pp += 0.5 * exp_rho + p_equilibrium * (1.0 - exp_rho);
                            }
                            likelihood *= pp;
                            ph1 = ph2 = 99;
                        }
                    }
                }
                if (ph1 < 99) {
                    double mutprob = (s_mp[di[0] * PF_BS] + s_mp[di[1] * PF_BS]) * 0.5;
                    double pp = 0;
                    for (int r = 0; r < 2; r++)
                        pp += 0.5 * (mutprob + (1.0 - mutprob) * p_equilibrium *
                                     (1.0 - fastexp_approx(-rhoprime_c * (r == 0 ? 1.0 : 0.5) * l_mean * fed)));
                    likelihood *= pp;
                }
            }
        }
        if (A.la_split[row] > -1 && A.apf >= 3) {
            double rate_of_change = Ltree * recomb_rate / 2;
            double p_nochange = fastexp_approx(-rate_of_change * A.la_split[row]);
            unsigned one_mask = 0, zero_mask = 0;
            for (int i = 0; i < n; ++i) {
                int v = A.la_salleles[(size_t)row * n + i];
                if (v == 1) one_mask |= 1u << i;
                if (v == 0) zero_mask |= 1u << i;
            }
            double p_splitdata = site_lik_lane(ln, one_mask, zero_mask, false, m.t0 + threadIdx.x, m.t1 + threadIdx.x);
            int k = A.la_sk[row];
            double p_correct_split = k / double(4.0 * n * n);
            if (A.apf == 4) {
                double nCk = 1.0;
                for (int i = 1; i <= k; i++) nCk *= (n - i + 1) / double(i);
                p_correct_split = 1.0 / nCk;
            }
            double split_branch_length = k * A.la_mean_tbl / (2 * n * (0.577 * dlog((double)n)));
            double pp = p_nochange * p_splitdata + (1.0 - p_nochange) * p_correct_split * mut_rate * split_branch_length;
            likelihood *= pp;
        }
        // removeLookaheadLikelihood + includeLookaheadLikelihood (particle.hpp:182, particle.cpp:614-616)
        w_pilot = st.w_pilot[p];
        w_pilot /= st.lookahead[p];
        double la_w = 1.0;
        la_w *= likelihood;
        w_pilot *= likelihood;
        st.lookahead[p] = la_w;
        st.w_pilot[p] = w_pilot;
    }
    double sq = wave_tree_sum(w_pilot * w_pilot);
    double sc = wave_hs_scan(w_pilot, lane);
    double scm = wave_max_scan_d(sc, lane);
    long long chunk = p >> 6;
    if (active) { A.scan1[p] = sc; A.scan1m[p] = scm; }
    if (lane == 63 && chunk < (A.Np + 63) / 64) {
        A.chunk_sq[chunk] = sq;
        A.chunk_pil[chunk] = sc;
        A.chunk_mx1[chunk] = scm;
    }
}

// calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166): one prior tree per lane (Philox stream 3); reports
// every sample's coalescent parent height and the tree length
__global__ __launch_bounds__(PF_BS) void k_tbl(KArgs A, unsigned long long seed, long long rep0, long long nrep,
                                               double* out_h /* [n][nrep] */, double* out_len /* [nrep] */) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    load_model(A, m);
    __syncthreads();
    long long r = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (r >= nrep) return;
    const int n = A.n;
    Lane ln = make_lane(A, m, rep0 + r);
    ln.seed = seed;
    ln.stream = 3;
    ln.ebuf = -dlog(uni(ln));
    int root = 0;
    for (int i = 1; i < n; ++i) {
        int ni = i - 1;
        double tc = coalesce_up(ln, [&](int k) { return LS(ln, k); }, ni, i, 0.0);
        int pr = -1, ps = 0;
        int k = lineages_at(ln, ni, tc, -1, &pr, &ps);
        bool above_root = (ni == 0) || (tc >= LS(ln, ni - 1));
        int kk = above_root ? 1 : k;
        double u = uni(ln);
        int idx = min((int)(u * (double)kk), kk - 1);
        if (above_root) insert_node(ln, ni, tc, i, -1, 0, root);
        else { lineages_at(ln, ni, tc, idx, &pr, &ps); insert_node(ln, ni, tc, i, pr, ps, root); }
        root = n + ni;
    }
    out_len[r] = tree_length(ln, n);
    for (int i = 0; i < n; ++i) {
        int pr = -1;
        for (int k = 0; k < n - 1 && pr < 0; ++k)
            if (LC(ln, k, 0) == i || LC(ln, k, 1) == i) pr = k;
        out_h[(size_t)i * nrep + r] = LS(ln, pr);
    }
}

// exp_digamma(c)/c for every event count (particle.cpp:266-272), with the device's own exp / log
__global__ void k_vb_table(const double* counts, long long n, double* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = exp_digamma(counts[i]) / counts[i];
}

// ------------------------------------------------------------------ unit-test kernels
__global__ void k_test_math(const double* x, long long n, double* oe, double* ol, double* of) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    oe[i] = dexp(x[i]);
    ol[i] = x[i] > 0 ? dlog(x[i]) : 0.0;
    of[i] = fastexp(x[i]);
}
__global__ void k_test_div(const double* a, const double* b, long long n, double* o) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] / b[i];
}
__global__ void k_test_uniform(unsigned long long seed, unsigned slot, unsigned stream, unsigned long long first, long long n,
                               double* o) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = philox_uniform(seed, slot, stream, first + (unsigned long long)i);
}

// ------------------------------------------------------------------ k_simulate: synthetic data on the device
// The `.seg` producer next to the path (the reference shells out to scrm and converts its
// output, populationmodels.py:440-577).  One lane = one independent chromosome chunk: a prior tree, then along the
// sequence the same SMC' transition the filter simulates (genealogy_update), and between recombinations mutations
// dropped as a Poisson process of rate mu * tree length, each on a branch drawn in proportion to its length
// (sample_point) -- the carriers are the samples below it.  Output per chunk: site positions (continuous, ascending)
// and carrier masks; the host rounds them to bases and writes rows.  Its own Philox stream (3).
__global__ __launch_bounds__(PF_BS) void k_simulate(KArgs A, unsigned long long seed, int nchunks, long long max_sites,
                                                    double* pos_out, unsigned* mask_out, long long* n_out) {
    extern __shared__ double smem[];
    Smem m = carve(smem, A.n, A.E);
    load_model(A, m);
    __syncthreads();
    const long long r = (long long)blockIdx.x * PF_BS + threadIdx.x;
    if (r >= nchunks) return;
    const int n = A.n;
    Lane ln = make_lane(A, m, r);
    ln.seed = seed;
    ln.stream = 3;
    ln.ebuf = -dlog(uni(ln));
    int root = 0;
    for (int i = 1; i < n; ++i) {            // Forest::buildInitialTree: the leaves join one at a time (as k_calibrate)
        int ni = i - 1;
        double tc = coalesce_up(ln, [&](int k) { return LS(ln, k); }, ni, i, 0.0);
        int pr = -1, ps = 0;
        int k = lineages_at(ln, ni, tc, -1, &pr, &ps);
        bool above_root = (ni == 0) || (tc >= LS(ln, ni - 1));
        int kk = above_root ? 1 : k;
        double u = uni(ln);
        int idx = min((int)(u * (double)kk), kk - 1);
        if (above_root) insert_node(ln, ni, tc, i, -1, 0, root);
        else { lineages_at(ln, ni, tc, idx, &pr, &ps); insert_node(ln, ni, tc, i, pr, ps, root); }
        root = n + ni;
    }
    ln.Ltree = tree_length(ln, n);
    double* tmp = m.t0 + threadIdx.x;        // per-lane LDS column for the descendant masks
    double* pos = pos_out + (size_t)r * max_sites;
    unsigned* msk = mask_out + (size_t)r * max_sites;
    long long ns = 0;
    double x = 0.0;
    double next_rec = sample_next_base(ln, 0.0);
    double next_mut = x + (-dlog(uni(ln))) / (A.mu * ln.Ltree);
    bool overflow = false;
    while (x < A.L) {
        if (next_mut < next_rec && next_mut < A.L) {
            // a mutation on the current tree: a uniform point of the tree picks the branch
            int rp = 0, sb = 0;
            double h;
            ln.uqn = 0;
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
        double h, tc, sp;
        bool changed;
        genealogy_update(ln, &h, &tc, &sp, &changed);
        ln.uqn = 0;                           // what is left of the update's uniforms is not reused
        next_rec = sample_next_base(ln, x);
        next_mut = x + (-dlog(uni(ln))) / (A.mu * ln.Ltree);     // memoryless: redrawn under the new tree length
    }
    n_out[r] = overflow ? -ns : ns;
}

// ------------------------------------------------------------------ host side
static thread_local std::string g_err;
const char* pf_last_error(void) { return g_err.c_str(); }

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

struct pf_handle {
    int device = 0;
    hipStream_t stream = nullptr;      // filter stream: k_extend, k_decide, k_resample
    hipStream_t cstream = nullptr;     // counting stream: k_count, k_ledger (off the critical path)
    std::vector<hipEvent_t> sync_ev;   // ring of events ordering the two streams
    size_t sync_next = 0;
    hipEvent_t ev_dec = nullptr, ev_cnt = nullptr;   // last decide / last count+ledger
    KArgs A;
    std::vector<void*> allocs;
    std::vector<double> h_lags, h_counted_to;
    double h_L = 0;
    std::vector<double> h_seg_start, h_seg_len;
    long long n_segs = 0;
    long long seg_done = 0;
    int E = 0, n = 0, P = 1;
    long long Np = 0;
    int nblocks = 0;
    size_t smem = 0, smem_la = 0;
    int max_trace_events = 0;
    bool finished = false;
    bool fin_pending = false;     // k_count partials not yet folded into the totals
    Windows step_windows;         // windows of the step being processed
    bool force_lds = false;       // pf_params.debug & PF_DEBUG_FORCE_LDS: use the LDS-tree kernel for every n (testing)
    bool no_fuse = false;         // PF_DEBUG_NO_FUSE: always run k_resample as its own kernel (testing)
    bool no_count = false;        // PF_DEBUG_NO_COUNT: no lagged counting, no ledger upkeep (profiling)
    bool two_launch_rows = false; // PF_DEBUG_TWO_LAUNCH: the round-1 row pipeline (k_row + k_decide_ledger) instead of k_pipe
    bool pipe = false;            // the single-launch pipeline applies (one population, n <= 8; rings allocated)
    size_t smem_pipe = 0;
    int ncw = 0;                  // count workgroups per epoch in the row pipeline (pf_params.count_wgs): the most a column gets
    int ledger_wgs = 192;         // workgroups per step that re-base the older generations' run lists after a resampling (beside one per particle block)
    int workers = 0;              // pf_params.count_workers: workgroups per step that take the ledger and count items off a queue (0: one workgroup per item)
    std::vector<int> cw_off;      // [E + 1] first count workgroup of the j-th column, oldest epoch first (KArgs::cw_off)
    bool flag_handoff = false;    // PF_DEBUG_FLAG_HANDOFF: one population, rows as extend / draw launches that alternate between two streams and
                                  // bookkeeping / ledger / count launches on the counting stream, ordered by counters in memory instead of events
    hipStream_t stream2 = nullptr;
    bool sweep_handoff = false;   // (argument of sweep_table: the table it builds is for run_sweep_flags)
    bool sweep_split2 = false;    // (argument of sweep_table: ... for run_sweep_split)
    bool split_many = false;      // a step as two launches (run_sweep_split; not with PF_DEBUG_ONE_LAUNCH)
    int split_batch = 0;          // ... the second launches enqueued in batches of this many steps behind one completion signal (0: chosen by the runner)
    unsigned long long* d_trace = nullptr;   // pf_set_wg_trace (leader of a pf_run_many call): four words per workgroup and step
    size_t trace_words = 0;
    int trace_t0 = 0, trace_n = 0, trace_stride = 0, trace_grid[3] = {0, 0, 0};
    bool split_roles = false;     // PF_DEBUG_SPLIT_ROLES: one population too runs the extend role and the other roles as two launches on two streams
    bool pipe_mp = false;         // structured models on the row pipeline: extend launches on the filter stream, the other roles on the counting stream
    size_t smem_sweep_x = 0;
    std::vector<hipEvent_t> ev_x, ev_blc;   // completion of the last 16 extend / bookkeeping-ledger-count launches
    bool no_spec_stage = false;   // PF_DEBUG_NO_SPEC_STAGE
    bool use_k_pipe = false;      // PF_DEBUG_K_PIPE: rows through k_pipe (argument block by value, one chunk per launch) instead of k_sweep
    SweepChunk* d_sweep = nullptr; // device table of the chunks this handle leads through k_sweep
    int d_sweep_cap = 0;
    std::vector<SweepChunk> h_sweep;
    // timing
    int timing_period = 0;
    struct Span { hipEvent_t a, b; int k; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    double span_overhead_ms = 0;  // what a pair of HIP events brackets when nothing is between them (subtracted from every span)
    double k_ms[4] = {0, 0, 0, 0};
    long long k_launches[4] = {0, 0, 0, 0};
    long long k_timed[4] = {0, 0, 0, 0};
};

template <class T>
static int dalloc(pf_handle* h, T** p, size_t count) {
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), h->stream));
    h->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}

// Bucket table of r_search_lut for an ascending table tab[0..E) with tab[0] = 0: lut[j] = the largest e with tab[e] <= the
// lower edge of bucket j, bucket j = the doubles whose upper sixteen bits are kbase + j (bucket 0 also takes everything
// below, the last bucket everything above).  Returns false when the table spans more buckets than PF_LUT_N, when a bucket
// holds more than two entries (the two probes of the device would not finish the search) or when the probes could leave
// the padded table.
static bool lut_build(const double* tab, int E, unsigned char* lut, int* kbase) {
    auto key = [](double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(b >> 48); };
    auto edge = [](int k) { uint64_t b = (uint64_t)k << 48; double v; memcpy(&v, &b, 8); return v; };
    if (E + 2 > PF_EPAD) return false;
    for (int e = 0; e < E; ++e) if (!(tab[e] >= 0.0) || !std::isfinite(tab[e]) || (e > 0 && tab[e] < tab[e - 1])) return false;
    if (E == 1 || !(tab[E - 1] > 0.0)) { memset(lut, 0, PF_LUT_N); *kbase = 0x3ff0; return E == 1; }
    int first = 1;
    while (first < E && tab[first] == 0.0) ++first;              // entries equal to tab[0] (an epoch of zero intensity at the start)
    const int kb = key(tab[first]) - 1;
    if (kb < 1 || key(tab[E - 1]) > kb + PF_LUT_N - 2) return false;
    *kbase = kb;
    for (int j = 0; j < PF_LUT_N; ++j) {
        const double lo = j == 0 ? 0.0 : edge(kb + j);
        int e = 0;
        while (e + 1 < E && tab[e + 1] <= lo) ++e;
        lut[j] = (unsigned char)e;
        if (j + 1 < PF_LUT_N) {
            const double hi = edge(kb + j + 1);
            int inside = 0;
            for (int q = e + 1; q < E && tab[q] < hi; ++q) ++inside;
            if (inside > 2) return false;
        } else if (e != E - 1) return false;
    }
    return true;
}

// cumulative coalescence intensity at the epoch starts, in exactly this order of operations on both sides of the
// parity tests: Hc[0] = 0, Hc[e+1] = Hc[e] + (T[e+1] - T[e]) * inv2N[e]
static std::vector<double> cumulative_intensity(const double* T, const std::vector<double>& inv2N, int E) {
    std::vector<double> H(E, 0.0);
    for (int e = 0; e + 1 < E; ++e) H[e + 1] = H[e] + (T[e + 1] - T[e]) * inv2N[e];
    return H;
}

int pf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pf_destroy(pf_handle* h);

// per-epoch, per-population tables of a structured model (scrm Model::population_size / migration_rate /
// single_mig_pop), prepared exactly as the oracle's fill_model does
struct MpTables {
    std::vector<double> inv2Np, mrate, mtot;
    std::vector<double> cum_coal, cum_mig;     // [E*P] integrals of 1/2N_p and of the total emigration rate of p from 0 to the epoch start
    std::vector<double> next_join;             // [E] start of the next epoch after e that moves a whole population (-ej), +inf if none
    std::vector<int> jmap, spop, next_join_epoch;
};
static int build_mp_tables(const pf_model* m, MpTables& t) {
    const int E = m->n_epochs, P = m->n_pops, n = m->nsam;
    t.inv2Np.resize((size_t)E * P);
    for (int i = 0; i < E * P; ++i) t.inv2Np[i] = 1.0 / (2.0 * m->pop_sizes[i]);
    t.mrate.assign((size_t)E * P * P, 0.0);
    if (m->mig_rates) t.mrate.assign(m->mig_rates, m->mig_rates + (size_t)E * P * P);
    t.mtot.assign((size_t)E * P, 0.0);
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < P; ++a) {
            double sum = 0.0;
            for (int b = 0; b < P; ++b) if (b != a) sum += t.mrate[((size_t)e * P + a) * P + b];
            t.mtot[(size_t)e * P + a] = sum;
        }
    t.jmap.resize((size_t)E * P);
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < P; ++a) {
            int cur = a;
            for (int step = 0; step <= P && m->single_mig; ++step) {
                int nxt = cur;
                for (int b = 0; b < P; ++b) {
                    double pr = m->single_mig[((size_t)e * P + cur) * P + b];
                    if (pr != 0.0 && pr != 1.0) { g_err = "partial single migration events (-es/-eps style) are not supported"; return -1; }
                    if (pr == 1.0 && b != cur) { nxt = b; break; }
                }
                if (nxt == cur) break;
                if (step == P) { g_err = "Cycle detected when moving individuals between populations"; return -1; }
                cur = nxt;
            }
            t.jmap[(size_t)e * P + a] = cur;
        }
    // Tables of the structured walk (pf_mp.h mp_coalesce): the hazard of a lineage over a stretch that spans epochs is
    // a difference of these; accumulated in this order, which the oracle restates.
    t.cum_coal.assign((size_t)E * P, 0.0);
    t.cum_mig.assign((size_t)E * P, 0.0);
    for (int e = 0; e + 1 < E; ++e)
        for (int a = 0; a < P; ++a) {
            const double dt = m->change_times[e + 1] - m->change_times[e];
            t.cum_coal[(size_t)(e + 1) * P + a] = t.cum_coal[(size_t)e * P + a] + dt * t.inv2Np[(size_t)e * P + a];
            t.cum_mig[(size_t)(e + 1) * P + a] = t.cum_mig[(size_t)e * P + a] + dt * t.mtot[(size_t)e * P + a];
        }
    t.next_join.assign(E, INFINITY);
    t.next_join_epoch.assign(E, E);
    for (int e = E - 2; e >= 0; --e) {
        bool moves = false;
        for (int a = 0; a < P; ++a) moves |= t.jmap[(size_t)(e + 1) * P + a] != a;
        t.next_join[e] = moves ? m->change_times[e + 1] : t.next_join[e + 1];
        t.next_join_epoch[e] = moves ? e + 1 : t.next_join_epoch[e + 1];
    }
    t.spop.assign(n, 0);
    if (m->sample_pops) t.spop.assign(m->sample_pops, m->sample_pops + n);
    for (int v : t.spop) if (v < 0 || v >= P) { g_err = "sample population out of range"; return -1; }
    return 0;
}

static int upload_mp_tables(const MpTables& t, KArgs& A, std::vector<void*>& allocs) {
    double *d1, *d2, *d3, *d4, *d5, *d6; int *i1, *i2, *i3;
    HIPCHK(hipMalloc(&i3, t.next_join_epoch.size() * 4)); allocs.push_back(i3);
    HIPCHK(hipMemcpy(i3, t.next_join_epoch.data(), t.next_join_epoch.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&d4, t.cum_coal.size() * 8)); allocs.push_back(d4);
    HIPCHK(hipMalloc(&d5, t.cum_mig.size() * 8)); allocs.push_back(d5);
    HIPCHK(hipMalloc(&d6, t.next_join.size() * 8)); allocs.push_back(d6);
    HIPCHK(hipMemcpy(d4, t.cum_coal.data(), t.cum_coal.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d5, t.cum_mig.data(), t.cum_mig.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d6, t.next_join.data(), t.next_join.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&d1, t.inv2Np.size() * 8)); allocs.push_back(d1);
    HIPCHK(hipMalloc(&d2, t.mrate.size() * 8)); allocs.push_back(d2);
    HIPCHK(hipMalloc(&d3, t.mtot.size() * 8)); allocs.push_back(d3);
    HIPCHK(hipMalloc(&i1, t.jmap.size() * 4)); allocs.push_back(i1);
    HIPCHK(hipMalloc(&i2, t.spop.size() * 4)); allocs.push_back(i2);
    HIPCHK(hipMemcpy(d1, t.inv2Np.data(), t.inv2Np.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2, t.mrate.data(), t.mrate.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d3, t.mtot.data(), t.mtot.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(i1, t.jmap.data(), t.jmap.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(i2, t.spop.data(), t.spop.size() * 4, hipMemcpyHostToDevice));
    A.inv2Np = d1; A.mig_rate = d2; A.mig_tot = d3; A.join_map = i1; A.sample_pop = i2;
    A.cum_coal = d4; A.cum_mig = d5; A.next_join = d6; A.next_join_epoch = i3;
    return 0;
}

static pf_handle* create_impl(const pf_model* m, const pf_params* p, int device, long long log_cap, long long gen_cap) {
    auto fail = [&](const std::string& msg) -> pf_handle* { g_err = msg; return nullptr; };
    int ndev = pf_device_count();
    if (ndev <= 0) return fail("pf_create: no HIP device available (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("pf_create: device index out of range");
    if (m->n_pops < 1 || m->n_pops > PF_PMAX) return fail("pf_create: n_pops must be in 1..4");
    if (m->nsam < 2 || m->nsam > PF_NMAX) return fail("pf_create: nsam must be in 2..16");
    if (m->n_epochs < 1 || m->n_epochs > PF_EMAX) return fail("pf_create: n_epochs must be in 1..64");
    if (p->np < 1 || p->np > 262144) return fail("pf_create: np must be in 1..262144");
    if (m->n_bias_heights < 0 || m->n_bias_heights > PF_BIAS_MAX) return fail("pf_create: at most 8 bias heights are supported");
    if (m->n_bias_heights > 0 && (!m->bias_heights || !m->bias_strengths || !m->application_delays))
        return fail("pf_create: bias_heights, bias_strengths and application_delays must all be given");
    if (m->n_rate_segments > 0) {
        if (!m->rate_positions || !m->rate_values || !m->leaf_rel_rates || !m->application_delays)
            return fail("pf_create: rate_positions, rate_values, leaf_rel_rates and application_delays must all be given with a guide");
        if (m->rate_positions[0] != 0.0) return fail("pf_create: the recombination guide must start at position 0");
        for (int k = 0; k < m->n_rate_segments; ++k) {
            if (k && !(m->rate_positions[k] > m->rate_positions[k - 1])) return fail("pf_create: guide segment starts must increase");
            if (!(m->rate_values[k] > 0)) return fail("pf_create: guide recombination rates must be positive");
            for (int i = 0; i < m->nsam; ++i)
                if (!(m->leaf_rel_rates[(size_t)k * m->nsam + i] > 0)) return fail("pf_create: relative leaf rates of the guide must be positive");
        }
    }
    pf_handle* h = new pf_handle();
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete h; return fail("hipSetDevice failed"); }
    // (no stream priorities: a high-priority filter stream per handle gains nothing for one chunk and costs 30 % of the
    // throughput when several chunks share the device -- priority streams share fewer hardware queues)
    if ((p->debug & PF_DEBUG_SPLIT_ROLES) && (p->debug & PF_DEBUG_CU_MASK) && m->n_pops == 1) {
        // the extend launches on compute units of their own: the count workgroups of the other stream, which wait on memory
        // most of the time but share issue slots, LDS and the vector memory path with whatever sits beside them, stay off them
        int ncu = 0;
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        const int words = (ncu + 31) / 32;
        const int nx = std::min(ncu / 2, (int)((p->np + PF_BS - 1) / PF_BS) * 2 + 8);       // extend + draw workgroups
        std::vector<uint32_t> mx((size_t)words, 0u), mc((size_t)words, 0u);
        for (int i = 0; i < ncu; ++i) (i < nx ? mx : mc)[(size_t)(i >> 5)] |= 1u << (i & 31);
        if (hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)words, mx.data()) != hipSuccess ||
            hipExtStreamCreateWithCUMask(&h->cstream, (uint32_t)words, mc.data()) != hipSuccess) { delete h; return fail("hipExtStreamCreateWithCUMask failed"); }
    } else {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail("hipStreamCreate failed"); }
    if (hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail("hipStreamCreate failed"); }
    }
    h->sync_ev.resize(512);
    for (auto& e : h->sync_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { delete h; return fail("hipEventCreate failed"); }
    const int E = m->n_epochs, n = m->nsam;
    const long long Np = p->np;
    const int P = m->n_pops;
    h->E = E; h->n = n; h->Np = Np; h->P = P;
    h->nblocks = (int)((Np + PF_BS - 1) / PF_BS);
    const int mcap = p->mig_cap > 0 ? p->mig_cap : PF_MMAX;
    if (mcap > 4096) { delete h; return fail("pf_create: mig_cap out of range"); }
    h->smem = P > 1 ? pf_mp_smem_bytes(n, E, P, mcap) : smem_bytes(n, E);
    h->max_trace_events = std::max(0, p->max_trace_events);
    if (p->flags & 2) {
        // -arg: the parent table of every resampling is kept (it is the ancestry the tree dump walks back through)
        if (P > 1 && (n > 8 || (p->debug & PF_DEBUG_FORCE_LDS))) {
            delete h;
            return fail("pf_create: tree recording (-arg) with several populations needs nsam <= 8 (register-tree kernel)");
        }
        if (gen_cap > 0x7fffffffLL / 2) gen_cap = 0x7fffffffLL / 2;
        h->max_trace_events = (int)gen_cap;
    }
    h->force_lds = (p->debug & PF_DEBUG_FORCE_LDS) != 0;
    h->no_fuse = (p->debug & PF_DEBUG_NO_FUSE) != 0;
    h->no_count = (p->debug & PF_DEBUG_NO_COUNT) != 0;
    h->two_launch_rows = (p->debug & PF_DEBUG_TWO_LAUNCH) != 0;
    h->use_k_pipe = (p->debug & PF_DEBUG_K_PIPE) != 0;
    h->no_spec_stage = (p->debug & PF_DEBUG_NO_SPEC_STAGE) != 0;
    h->split_roles = (p->debug & PF_DEBUG_SPLIT_ROLES) != 0;
    h->split_batch = (p->debug >> 16) & 15;       // (bits 16-19 of debug: tuning experiments; 0 = four with several chunks, one with one)
    // one population, at most four haplotypes, no focused sampling, no -arg: a step is two launches (run_sweep_split) unless the switch says one
    h->split_many = !(p->debug & PF_DEBUG_ONE_LAUNCH) && P == 1 && n <= 4 && m->n_bias_heights == 0 && m->n_rate_segments == 0 && !(p->flags & 2) &&
                    !(p->debug & (PF_DEBUG_SPLIT_ROLES | PF_DEBUG_FLAG_HANDOFF | PF_DEBUG_K_PIPE | PF_DEBUG_TWO_LAUNCH | PF_DEBUG_NO_FUSE));
    // (several chunks per GPU -- the caller set count_wgs -- : on the rows that do not resample, nine in ten, these workgroups find nothing to do and
    // leave after 2 us of a slot each; with 32 instead of 192 eight chunks gain 3 %, one chunk is the same either way: 28.7 us per row)
    h->ledger_wgs = p->count_wgs > 0 ? 32 : std::max(16, std::min(PF_LEDGER_BLOCKS, 192));
    // (instantiated for the headline shape only: at most four haplotypes, no focused sampling or guide, no tree dump)
    h->flag_handoff = (p->debug & PF_DEBUG_FLAG_HANDOFF) != 0 && P == 1 && n <= 4 && m->n_bias_heights == 0 && m->n_rate_segments == 0 && !(p->flags & 2);
    if (h->flag_handoff && hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess) { delete h; return fail("hipStreamCreate failed"); }
    h->h_lags.assign(m->lags, m->lags + E);
    h->h_counted_to.assign(E, 0.0);
    h->h_L = m->loci_length;
    KArgs& A = h->A;
    memset(&A, 0, sizeof(A));
    A.E = E; A.n = n; A.flags = m->flags | (h->no_spec_stage ? 256 : 0) | ((p->debug & PF_DEBUG_COUNT_YOUNG_FIRST) ? 512 : 0);
    A.flags |= p->debug & (7 << 20);        // profiling probes of the count role (bits 20, 21: the sums are then wrong on purpose)
    A.L = m->loci_length; A.mu = m->mutation_rate; A.rho = m->recombination_rate;
    A.Np = Np;
    A.mcap = mcap;
    A.ess_threshold = (double)Np * p->ess_fraction;
    A.seed = p->seed;
    int rc = 0;
    double *dT, *dI, *dlag, *dHc; int* dRF;
    rc |= dalloc(h, &dT, E); rc |= dalloc(h, &dI, E); rc |= dalloc(h, &dlag, E); rc |= dalloc(h, &dRF, E); rc |= dalloc(h, &dHc, E);
    if (rc) { pf_destroy(h); return nullptr; }
    std::vector<double> inv2N(E);
    for (int e = 0; e < E; ++e) inv2N[e] = 1.0 / (2.0 * m->pop_sizes[(size_t)e * P]);
    A.P = P;
    {
        const int PT = P <= 2 ? P : PF_PMAX;     // k_count is instantiated for 1, 2 and PF_PMAX populations
        A.ncol = PT == 1 ? 6 : 3 * PT + 3 + PT * PT + 2 * PT;
    }
    if (P > 1) {
        MpTables tb;
        if (build_mp_tables(m, tb) || upload_mp_tables(tb, A, h->allocs)) { pf_destroy(h); return nullptr; }
    }
    if (m->vb_coal_counts) {
        const size_t nc = (size_t)E * P, nm = (size_t)E * P * P;
        double *cin, *cout_, *mout;
        if (dalloc(h, &cin, nc + nm) || dalloc(h, &cout_, nc) || dalloc(h, &mout, nm)) { pf_destroy(h); return nullptr; }
        std::vector<double> cnt(nc + nm, 1e10);
        for (size_t i = 0; i < nc; ++i) cnt[i] = m->vb_coal_counts[i];
        if (m->vb_mig_counts) for (size_t i = 0; i < nm; ++i) cnt[nc + i] = m->vb_mig_counts[i];
        for (double v : cnt) if (!(v > 0)) { pf_destroy(h); return fail("pf_create: variational-Bayes event counts must be positive"); }
        hipMemcpyAsync(cin, cnt.data(), cnt.size() * 8, hipMemcpyHostToDevice, h->stream);
        hipLaunchKernelGGL(k_vb_table, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream, cin, (long long)nc, cout_);
        hipLaunchKernelGGL(k_vb_table, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, h->stream, cin + nc, (long long)nm, mout);
        hipStreamSynchronize(h->stream);
        A.vb_coal = cout_; A.vb_mig = mout;
    }
    hipMemcpyAsync(dT, m->change_times, E * 8, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(dI, inv2N.data(), E * 8, hipMemcpyHostToDevice, h->stream);
    const std::vector<double> Hc = cumulative_intensity(m->change_times, inv2N, E);
    hipMemcpyAsync(dHc, Hc.data(), E * 8, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(dlag, m->lags, E * 8, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(dRF, m->record_flags, E * 4, hipMemcpyHostToDevice, h->stream);
    hipStreamSynchronize(h->stream);
    A.T = dT; A.inv2N = dI; A.Hc = dHc; A.lags = dlag; A.recflags = dRF;
    A.lut = nullptr; A.lut_kbT = 0; A.lut_kbH = 0;
    if (!(p->debug & PF_DEBUG_NO_SEARCH_LUT)) {
        unsigned char lut[2 * PF_LUT_N];
        if (lut_build(m->change_times, E, lut, &A.lut_kbT) && lut_build(Hc.data(), E, lut + PF_LUT_N, &A.lut_kbH)) {
            unsigned char* dl;
            rc |= dalloc(h, &dl, 2 * PF_LUT_N);
            // on the handle's stream, behind the memset of dalloc (a plain hipMemcpy is not ordered with a non-blocking stream)
            if (!rc) { hipMemcpyAsync(dl, lut, sizeof(lut), hipMemcpyHostToDevice, h->stream); hipStreamSynchronize(h->stream); A.lut = dl; }
        }
    }
    A.n_bias = m->n_bias_heights;
    A.delay_type = m->delay_type;
    A.dcap = p->delay_cap > 0 ? p->delay_cap : PF_DCAP_DEFAULT;
    A.delay_evict = (p->flags & 4) ? 1 : 0;
    for (int k = 0; k < PF_BIAS_MAX + 2; ++k) A.bias_H[k] = HUGE_VAL;
    for (int k = 0; k < PF_BIAS_MAX + 1; ++k) A.bias_S[k] = 1.0;
    A.bias_H[0] = 0.0;
    if (A.n_bias > 0 || m->n_rate_segments > 0) {
        for (int k = 0; k < A.n_bias; ++k) A.bias_H[k + 1] = m->bias_heights[k];
        for (int k = 0; k <= A.n_bias && A.n_bias > 0; ++k) A.bias_S[k] = m->bias_strengths[k];
        double* dad;
        if (dalloc(h, &dad, E)) { pf_destroy(h); return nullptr; }
        // on the handle's stream, behind the memset of dalloc: a plain hipMemcpy is not ordered with a non-blocking stream, and
        // the memset, queued behind those of the state arrays, could land after it and leave every delay at zero
        hipMemcpyAsync(dad, m->application_delays, E * 8, hipMemcpyHostToDevice, h->stream);
        hipStreamSynchronize(h->stream);
        A.app_delays = dad;
    }
    h->pipe = P == 1 && n <= 8 && Np <= 131072;      // beyond that the decision tables outgrow the default dynamic LDS
    // structured models with the tree in registers: the same pipeline, the extend role as its own launch (run_sweep_mp)
    h->pipe_mp = P > 1 && n <= 8 && Np <= 131072 && !(p->flags & 2) && !(p->debug & (PF_DEBUG_FORCE_LDS | PF_DEBUG_NO_FUSE | PF_DEBUG_K_PIPE));
    A.blk_gran = h->pipe_mp ? 4 : 1;
    {
        const size_t K = (h->pipe || h->pipe_mp) ? PF_RING : 2;               // copies of the particle state (KArgs::st0)
        A.nslots = (int)K;
        DState& st = A.st0;
        rc |= dalloc(h, &st.S, K * (size_t)(n - 1) * Np);
        rc |= dalloc(h, &st.C, K * (size_t)2 * (n - 1) * Np);
        rc |= dalloc(h, &st.w_post, K * Np);
        rc |= dalloc(h, &st.w_pilot, K * Np);
        rc |= dalloc(h, &st.next_base, K * Np);
        rc |= dalloc(h, &st.x_mark, K * Np);
        rc |= dalloc(h, &st.Ltree, K * Np);
        rc |= dalloc(h, &st.mark_limit, K * Np);
        if (P > 1) {
            rc |= dalloc(h, &st.Pn, K * (size_t)(n - 1) * Np);
            rc |= dalloc(h, &st.nm, K * Np);
            rc |= dalloc(h, &st.Mt, K * (size_t)A.mcap * Np);
            rc |= dalloc(h, &st.Mb, K * (size_t)A.mcap * Np);
            rc |= dalloc(h, &st.Mq, K * (size_t)A.mcap * Np);
        }
        if (m->n_rate_segments > 0) rc |= dalloc(h, &st.ridx, K * Np);
        if (m->n_bias_heights > 0 || m->n_rate_segments > 0) {
            rc |= dalloc(h, &st.total_delayed, K * Np);
            rc |= dalloc(h, &st.dcount, K * Np);
            rc |= dalloc(h, &st.dpos, K * (size_t)A.dcap * Np);
            rc |= dalloc(h, &st.dfac, K * (size_t)A.dcap * Np);
            rc |= dalloc(h, &st.ddelta, K * (size_t)A.dcap * Np);
            rc |= dalloc(h, &st.dk, K * (size_t)A.dcap * Np);
        }
    }
    if (p->flags & 1) {
        // 100-bp local recombination map (count.hpp:101-102, 115)
        A.lmap_bins = (long long)(m->loci_length / 100.0) + 4;
        rc |= dalloc(h, &A.lmap_opp, (size_t)A.lmap_bins);
        rc |= dalloc(h, &A.lmap_cnt, (size_t)(n + 2) * A.lmap_bins);
    }
    if (m->n_rate_segments > 0) {
        // RecombinationBias::set_model_rates (pfparam.hpp:199-211): the segments that start inside the locus
        int K = 0;
        while (K < m->n_rate_segments && m->rate_positions[K] < m->loci_length) ++K;
        double *gp, *gr, *gl;
        rc |= dalloc(h, &gp, K); rc |= dalloc(h, &gr, K); rc |= dalloc(h, &gl, (size_t)K * n);
        if (!rc) {
            hipMemcpyAsync(gp, m->rate_positions, (size_t)K * 8, hipMemcpyHostToDevice, h->stream);
            hipMemcpyAsync(gr, m->rate_values, (size_t)K * 8, hipMemcpyHostToDevice, h->stream);
            hipMemcpyAsync(gl, m->leaf_rel_rates, (size_t)K * n * 8, hipMemcpyHostToDevice, h->stream);
            hipStreamSynchronize(h->stream);
        }
        A.g_K = K; A.g_pos = gp; A.g_rho = gr; A.g_leaf = gl;
    }
    rc |= dalloc(h, &A.rng_ctr, Np);
    A.dt_tab = nullptr; A.dt_filled = nullptr; A.dt_ctr = nullptr;
    if (h->pipe && !h->use_k_pipe && !(p->debug & PF_DEBUG_NO_DRAW_TABLE)) {
        // draw table of k_sweep (draw_role): 512 bytes per slot
        rc |= dalloc(h, &A.dt_tab, (size_t)2 * PF_DRAW_RING * Np);
        rc |= dalloc(h, &A.dt_filled, (size_t)2 * Np);
        rc |= dalloc(h, &A.dt_ctr, (size_t)2 * Np);
    }
    rc |= dalloc(h, &A.ebuf, Np);
    rc |= dalloc(h, &A.widx, Np);
    A.cap = (unsigned)log_cap;
    A.rec_trees = (p->flags & 2) ? 1 : 0;
    A.RS = 5 + (n - 1);
    A.Gcap = (int)gen_cap;
    rc |= dalloc(h, &A.log, (size_t)Np * A.cap * A.RS);
    if (P > 1) {
        // a genealogy update leaves one piece per (epoch, population) stretch of its path: a handful per record
        A.pcap = (unsigned)(p->piece_cap > 0 ? p->piece_cap : 4 * log_cap);
        rc |= dalloc(h, &A.plog, (size_t)Np * A.pcap * 3);
        rc |= dalloc(h, &A.pidx, Np);
    }
    rc |= dalloc(h, &A.gstart, (size_t)A.Gcap * Np);
    rc |= dalloc(h, &A.lo, (size_t)A.Gcap * (Np + 1));
    rc |= dalloc(h, &A.gen_x0, A.Gcap);
    rc |= dalloc(h, &A.parent, Np);
    rc |= dalloc(h, &A.blkcnt2[0], (size_t)h->nblocks); rc |= dalloc(h, &A.blkcnt2[1], (size_t)h->nblocks);
    rc |= dalloc(h, &A.run_st, (size_t)A.Gcap * Np);
    rc |= dalloc(h, &A.run_anc, (size_t)A.Gcap * Np);
    rc |= dalloc(h, &A.nruns, A.Gcap);
    const size_t nc = (size_t)((Np + 63) / 64);
    A.nc = (int)nc;
    if (h->pipe || h->pipe_mp) {
        rc |= dalloc(h, &A.run_st2, (size_t)A.Gcap * Np);
        rc |= dalloc(h, &A.run_anc2, (size_t)A.Gcap * Np);
        rc |= dalloc(h, &A.nruns2, A.Gcap);
        rc |= dalloc(h, &A.rg_scan1, PF_RING * (size_t)Np); rc |= dalloc(h, &A.rg_scan1m, PF_RING * (size_t)Np);
        rc |= dalloc(h, &A.rg_scanp, PF_RING * (size_t)Np); rc |= dalloc(h, &A.rg_widx, PF_RING * (size_t)Np);
        rc |= dalloc(h, &A.rg_cpost, PF_RING * nc); rc |= dalloc(h, &A.rg_csq, PF_RING * nc); rc |= dalloc(h, &A.rg_cpil, PF_RING * nc);
        rc |= dalloc(h, &A.rg_cpp, PF_RING * nc); rc |= dalloc(h, &A.rg_cmx1, PF_RING * nc); rc |= dalloc(h, &A.rg_coffp, PF_RING * nc);
        rc |= dalloc(h, &A.rg_dpend, PF_RING * nc); rc |= dalloc(h, &A.rg_blkcnt, PF_RING * (size_t)h->nblocks * 4);
        h->smem_pipe = ((size_t)(2 * PF_EPAD + E + 2 * PF_BIAS_MAX + 3) + pipe_lds_doubles((int)nc)) * 8;
    }
    rc |= dalloc(h, &A.chunk_post, nc); rc |= dalloc(h, &A.chunk_sq, nc); rc |= dalloc(h, &A.chunk_pil, nc);
    rc |= dalloc(h, &A.scan1, Np); rc |= dalloc(h, &A.chunk_off, nc); rc |= dalloc(h, &A.l2scan, nc);
    rc |= dalloc(h, &A.scan1m, Np); rc |= dalloc(h, &A.chunk_mx1, nc);
    rc |= dalloc(h, &A.chunk_dpend, nc);
    rc |= dalloc(h, &A.chunk_pp, nc); rc |= dalloc(h, &A.l2scanp, nc);
    for (int b = 0; b < 2; ++b) {
        rc |= dalloc(h, &A.scanp2[b], Np); rc |= dalloc(h, &A.chunk_offp2[b], nc);
        rc |= dalloc(h, &A.snap_w[b], Np); rc |= dalloc(h, &A.snap_S[b], (size_t)(n - 1) * Np);
        rc |= dalloc(h, &A.snap_xm[b], Np); rc |= dalloc(h, &A.snap_ml[b], Np); rc |= dalloc(h, &A.snap_widx[b], Np);
    }
    // accumulators of the count workgroups per epoch (measured: four times as many count workgroups per epoch in the row
    // pipeline made a row 15 % slower -- the launch then holds 5 000 workgroups of 30 KB LDS each, four rounds of the chip)
    // per-epoch accumulators: one per count workgroup of a column (count_body)
    A.nbx = h->nblocks;
    // the row pipeline spreads an epoch's count tasks (the ancestor runs of the generations in its window: for the old epochs,
    // whose lag is a few rows, nearly one per particle) over this many workgroups; fewer is slower (C3 shape, one chunk: 40
    // workgroups 32.9 us per row, 16: 38.7, 8: 53.7, 4: 91.4, 2: 168 -- profiles/round3/count_wgs.md)
    h->ncw = p->count_wgs > 0 ? std::min<int>(p->count_wgs, h->nblocks) : h->nblocks;
    // (the queue has one kernel instance so far: one population, at most four haplotypes, no focused sampling, no -arg, single launch per step)
    h->workers = (h->pipe && !h->use_k_pipe && P == 1 && n <= 4 && !(m->n_bias_heights > 0 || m->n_rate_segments > 0) && !(p->flags & 2) &&
                  !(p->debug & (PF_DEBUG_SPLIT_ROLES | PF_DEBUG_FLAG_HANDOFF))) ? std::max(0, std::min(p->count_workers, 4096)) : 0;
    {
        // count workgroups per epoch column of the row pipeline: the tasks of an epoch are the live ancestors of the generations in its
        // window, about Np / (1 + depth), depth = generations between window and front, i.e. in proportion to its lag: a column whose
        // lag is a few rows long gets all h->ncw workgroups, the young epochs' columns, whose workgroups mostly found nothing to do
        // and left after their prologue, as few as two.  (What a workgroup finds it strides over: any number is correct.)
        h->cw_off.assign(E + 1, 0);
        for (int j = 0; j < E; ++j) {
            const double lag = m->lags[E - 1 - j];
            int w = (int)std::ceil((double)h->ncw * std::min(1.0, 2500.0 / std::max(lag, 1.0)));
            // (a higher minimum -- 4, 8, 12, 15 -- changes nothing: the long count workgroups of a step are not those of narrow columns)
            w = std::max(std::min(2, h->ncw), std::min(w, h->ncw));
            // (only when the caller set count_wgs, i.e. runs several chunks side by side and wants fewer workgroups: a single chunk
            // leaves most of the chip idle, and there every column is shortest with all the workgroups it can get)
            if ((p->debug & PF_DEBUG_COUNT_YOUNG_FIRST) || p->count_wgs <= 0) w = h->ncw;
            h->cw_off[j + 1] = h->cw_off[j] + w;
        }
        int* dcw = nullptr;
        if (h->pipe || h->pipe_mp) {
            rc |= dalloc(h, &dcw, E + 1);
            if (!rc) { hipMemcpyAsync(dcw, h->cw_off.data(), (size_t)(E + 1) * 4, hipMemcpyHostToDevice, h->stream); hipStreamSynchronize(h->stream); }
            // (with every column at full width the kernels compute column and workgroup from the index: no table)
            A.cw_off = ((p->debug & PF_DEBUG_COUNT_YOUNG_FIRST) || p->count_wgs <= 0) ? nullptr : dcw;
        }
    }
    rc |= dalloc(h, &A.totals, (size_t)A.ncol * E);
    rc |= dalloc(h, &A.partial, (size_t)E * A.nbx * A.ncol);
    A.max_trace_events = h->max_trace_events;
    rc |= dalloc(h, &A.ev_seg, (size_t)std::max(1, h->max_trace_events));
    rc |= dalloc(h, &A.ev_parents, (size_t)std::max(1, h->max_trace_events) * Np);
    rc |= dalloc(h, &A.ctrl, 1);
    if (rc) { pf_destroy(h); return nullptr; }
    if (P == 1 && h->smem > 64 * 1024) {
        hipFuncSetAttribute((const void*)k_extend, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
        hipFuncSetAttribute((const void*)k_init, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem);
    }
    // structured models: the LDS-tree kernels serve the prior tree and calibration for every n, the row kernel is the
    // register-tree one for n <= 8
    const bool lds_rows = P > 1 && (n > 8 || h->force_lds);
    if (h->smem > 160 * 1024 || (P > 1 && pf_mp_prepare(h->smem, A.mcap)) ||
        (P > 1 && !lds_rows && pf_mp_reg_smem_bytes(E, P, A.mcap) > 160 * 1024)) {
        pf_destroy(h);
        return fail(P > 1 ? "pf_create: the local-tree state (tree, epoch tables and pf_params.mig_cap migration events per lane) does not fit the LDS of one workgroup"
                          : "pf_create: the local-tree state does not fit the LDS of one workgroup");
    }
    if (h->pipe_mp) {
        h->smem_sweep_x = pf_mp_sweep_smem_bytes(E, P, A.mcap, A.nc);
        if (pf_mp_sweep_prepare(h->smem_sweep_x)) h->pipe_mp = false;        // the event lists and the decision tables do not fit together: the two-stream path
    }
    return h;
}

pf_handle* pf_create(const pf_model* m, const pf_params* p, int device) {
    // with -arg (flags bit 1) nothing may be overwritten: rings sized for a whole chunk by default
    const bool trees = p && (p->flags & 2);
    if (!m || !p) { g_err = "pf_create: null model or parameters"; return nullptr; }
    if (p->log_cap < 0 || p->gen_cap < 0 || p->piece_cap < 0 || p->log_cap > 0x7fffffffLL || p->piece_cap > 0x7fffffffLL) {
        g_err = "pf_create: ring capacities out of range";
        return nullptr;
    }
    if ((p->gen_cap > 0 && p->gen_cap < 4) || (p->log_cap > 0 && p->log_cap < 4)) {
        g_err = "pf_create: log_cap and gen_cap must be at least 4";
        return nullptr;
    }
    if (p->gen_cap > 0 && p->gen_cap < PF_RING + 4 && (m->n_pops > 1 || (p->debug & PF_DEBUG_SPLIT_ROLES))) {
        g_err = "pf_create: gen_cap must be at least 20 when the extend role runs ahead of the counts (structured models, PF_DEBUG_SPLIT_ROLES)";
        return nullptr;
    }
    return create_impl(m, p, device, p->log_cap > 0 ? p->log_cap : (trees ? 131072 : 16384),
                       p->gen_cap > 0 ? p->gen_cap : (trees ? 131072 : 8192));
}

void pf_destroy(pf_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->cstream) hipStreamSynchronize(h->cstream);
    for (void* p : h->allocs) hipFree(p);
    if (h->d_sweep) hipFree(h->d_sweep);
    if (h->d_trace) hipFree(h->d_trace);
    for (auto e : h->sync_ev) if (e) hipEventDestroy(e);
    for (auto e : h->ev_x) if (e) hipEventDestroy(e);
    for (auto e : h->ev_blc) if (e) hipEventDestroy(e);
    if (h->cstream) hipStreamDestroy(h->cstream);
    if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    for (auto& sp : h->spans) { hipEventDestroy(sp.a); hipEventDestroy(sp.b); }
    for (auto e : h->ev_pool) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_err = std::string(what) + ": " + hipGetErrorString(e); return -1; }
    return 0;
}

static int harvest_spans(pf_handle* h) {
    for (auto& sp : h->spans) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b));
        h->k_ms[sp.k] += std::max(0.0, (double)ms - h->span_overhead_ms);
        h->k_timed[sp.k] += 1;
        h->ev_pool.push_back(sp.a);
        h->ev_pool.push_back(sp.b);
    }
    h->spans.clear();
    return 0;
}

static hipEvent_t get_event(pf_handle* h) {
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

struct Timed {
    pf_handle* h; int k; bool on; hipEvent_t a, b; hipStream_t st;
    Timed(pf_handle* h_, int k_, bool on_, hipStream_t st_ = nullptr) : h(h_), k(k_), on(on_), st(st_ ? st_ : h_->stream) {
        h->k_launches[k] += 1;
        if (on) { a = get_event(h); b = get_event(h); hipEventRecord(a, st); }
    }
    ~Timed() {
        if (on) { hipEventRecord(b, st); h->spans.push_back({a, b, k}); }
    }
};

static hipEvent_t next_sync_event(pf_handle* h) {
    hipEvent_t e = h->sync_ev[h->sync_next];
    h->sync_next = (h->sync_next + 1) % h->sync_ev.size();
    return e;
}

int pf_sync(pf_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->cstream));
    if (h->fin_pending) {
        hipLaunchKernelGGL(k_count_fin, dim3(1), dim3(256), 0, h->stream, h->A);
        h->fin_pending = false;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (harvest_spans(h)) return -1;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    if (c.err) {
        const char* msg = "unknown device error";
        if (c.err == ERR_LOG_OVERFLOW) msg = "event log ring overflow (raise pf_params.log_cap)";
        if (c.err == ERR_GEN_OVERFLOW) msg = "generation ledger overflow (raise pf_params.gen_cap)";
        if (c.err == ERR_ZERO_PROB) msg = "Zero or negative probabilities";   /* pc.cpp:428-429 */
        if (c.err == ERR_MIG_OVERFLOW) msg = "too many migration events on one local tree";
        if (c.err == ERR_MP_INTERNAL) msg = "structured-model genealogy update: coalescence partners inconsistent";
        if (c.err == ERR_NO_COALESCENCE) msg = "No final coalescence event was sampled!";   /* particle.cpp:1383 */
        if (c.err == ERR_DELAY_OVERFLOW) msg = "delayed-factor store overflow (raise pf_params.delay_cap)";
        g_err = msg;
        return -2;
    }
    return 0;
}

int pf_init_prior(pf_handle* h, double initial_position) {
    HIPCHK(hipSetDevice(h->device));
    if (h->P > 1)
        pf_mp_launch_init(h->A, initial_position, h->smem, h->stream);
    else
        hipLaunchKernelGGL(k_init, dim3(h->nblocks), dim3(PF_BS), h->smem, h->stream, h->A, initial_position);
    if (check_launch("k_init")) return -1;
    if (h->A.lmap_opp) {
        HIPCHK(hipMemsetAsync(h->A.lmap_opp, 0, (size_t)h->A.lmap_bins * 8, h->stream));
        HIPCHK(hipMemsetAsync(h->A.lmap_cnt, 0, (size_t)(h->n + 2) * h->A.lmap_bins * 8, h->stream));
    }
    std::fill(h->h_counted_to.begin(), h->h_counted_to.end(), 0.0);
    HIPCHK(hipMemsetAsync(h->A.partial, 0, (size_t)h->E * h->A.nbx * h->A.ncol * 8, h->stream));
    h->fin_pending = false;
    h->ev_dec = nullptr; h->ev_cnt = nullptr;
    h->seg_done = 0;
    h->finished = false;
    return 0;
}

int pf_load_segments(pf_handle* h, const pf_segments* sg) {
    HIPCHK(hipSetDevice(h->device));
    const long long S = sg->n;
    double *ds, *dl; int8_t *dst, *dal; int* dlim;
    int rc = 0;
    rc |= dalloc(h, &ds, S); rc |= dalloc(h, &dl, S); rc |= dalloc(h, &dst, S);
    rc |= dalloc(h, &dal, (size_t)S * h->n); rc |= dalloc(h, &dlim, S);
    rc |= dalloc(h, &h->A.tr_T, S); rc |= dalloc(h, &h->A.tr_ess, S); rc |= dalloc(h, &h->A.tr_logl, S);
    rc |= dalloc(h, &h->A.tr_flag, S);
    if (rc) return -1;
    HIPCHK(hipMemcpyAsync(ds, sg->start, S * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dl, sg->length, S * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dst, sg->state, S, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dal, sg->alleles, (size_t)S * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dlim, sg->max_record_epoch, S * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->A.seg_start = ds; h->A.seg_len = dl; h->A.seg_state = dst; h->A.seg_alleles = dal; h->A.seg_limit = dlim;
    h->h_seg_start.assign(sg->start, sg->start + S);
    h->h_seg_len.assign(sg->length, sg->length + S);
    h->n_segs = S;
    return 0;
}

// the particle-independent window rule of extract_and_update_count (count.cpp:363-385)
static Windows host_windows(pf_handle* h, double current_base, bool end_data) {
    const int E = h->E;
    Windows W;
    memset(&W, 0, sizeof(W));
    W.first = E;
    W.end_data = end_data ? 1 : 0;
    for (int e = 0; e < E; ++e) {
        double lagging = end_data ? 0.0 : h->h_lags[e];
        double x_end = current_base - lagging;
        W.a[e] = h->h_counted_to[e];
        if ((x_end - h->h_counted_to[e]) < lagging * 0.1 && W.first > e) {
            W.b[e] = h->h_counted_to[e];
        } else {
            W.b[e] = x_end;
            W.first = std::min(W.first, e);
        }
    }
    for (int e = 0; e < E; ++e) h->h_counted_to[e] = W.b[e];
    return W;
}

static bool timing_on(pf_handle* h, long long s) { return h->timing_period > 0 && (s % h->timing_period) == 0; }

static Windows no_windows(pf_handle* h) {
    Windows W;
    memset(&W, 0, sizeof(W));
    W.first = h->E;
    for (int e = 0; e < h->E; ++e) { W.a[e] = h->h_counted_to[e]; W.b[e] = h->h_counted_to[e]; }
    return W;
}

// the register-tree kernels can complete the previous row while loading the particle (fused k_resample)
static bool extend_can_fuse(const pf_handle* h) {
    return h->P == 1 && h->n <= 8 && !h->force_lds && !h->no_fuse && h->A.apf == 0;
}

static int launch_extend(pf_handle* h, long long s, int fuse = 0) {
    const bool t = timing_on(h, s);
    {
        Timed tm(h, 0, t);
        const size_t smem_reg = (size_t)(2 * PF_EPAD + h->E + 2 * PF_BIAS_MAX + 3) * 8;
        const bool biased = h->A.n_bias > 0 || h->A.g_K > 0;
        if (h->P > 1)
            pf_mp_launch_extend(h->A, s, h->smem, h->stream, h->force_lds, fuse);
        else if (h->n <= 4 && biased && !h->force_lds)
        {
            if (h->A.rec_trees) hipLaunchKernelGGL((k_extend_reg<4, true, true>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
            else hipLaunchKernelGGL((k_extend_reg<4, true>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
        }
        else if (h->n <= 8 && biased && !h->force_lds)
        {
            if (h->A.rec_trees) hipLaunchKernelGGL((k_extend_reg<8, true, true>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
            else hipLaunchKernelGGL((k_extend_reg<8, true>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
        }
        else if (h->n <= 4 && !h->force_lds)
        {
            if (h->A.rec_trees) hipLaunchKernelGGL((k_extend_reg<4, false, true>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
            else hipLaunchKernelGGL((k_extend_reg<4, false>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
        }
        else if (h->n <= 8 && !h->force_lds)
        {
            if (h->A.rec_trees) hipLaunchKernelGGL((k_extend_reg<8, false, true>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
            else hipLaunchKernelGGL((k_extend_reg<8, false>), dim3(h->nblocks), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse);
        }
        else
            hipLaunchKernelGGL(k_extend, dim3(h->nblocks), dim3(PF_BS), h->smem, h->stream, h->A, s);
    }
    if (check_launch("k_extend")) return -1;
    if (h->A.apf > 0) {
        hipLaunchKernelGGL(k_lookahead, dim3(h->nblocks), dim3(PF_BS), h->smem_la, h->stream, h->A, s);
        return check_launch("k_lookahead");
    }
    return 0;
}

static int launch_decide(pf_handle* h, long long s, int mode, const Windows& W) {
    const bool t = timing_on(h, s);
    // k_decide rewrites what k_count / k_ledger read
    // (window generations, offspring tables): it must not start before the counting stream is done with them
    if (h->ev_cnt) hipStreamWaitEvent(h->stream, h->ev_cnt, 0);
    {
        Timed tm(h, 1, t);
        if (!h->no_count) {
            // the event the counting stream waits on is the kernel's own completion signal (no marker packet between
            // this kernel and the next row's extend)
            h->ev_dec = next_sync_event(h);
            hipExtLaunchKernelGGL(k_decide, dim3(h->nblocks + 1), dim3(PF_BS), 0, h->stream, nullptr, h->ev_dec, 0, h->A, s, mode, W, h->nblocks);
        } else {
            hipLaunchKernelGGL(k_decide, dim3(h->nblocks + 1), dim3(PF_BS), 0, h->stream, h->A, s, mode, W, h->nblocks);
        }
    }
    return check_launch("k_decide");
}

static int launch_count(pf_handle* h, long long s, const Windows& W) {
    if (h->no_count) return 0;
    const int first = W.first;
    if (first >= h->E) return 0;
    const bool t = timing_on(h, s);
    if (h->ev_dec) hipStreamWaitEvent(h->cstream, h->ev_dec, 0);
    {
        Timed tm(h, 2, t, h->cstream);
        const dim3 grid(h->nblocks, h->E - first), blk(PF_BS);
#define PF_LAUNCH_COUNT(NMV, PV) hipLaunchKernelGGL((k_count<NMV, PV>), grid, blk, 0, h->cstream, h->A, first, W)
        const int P = h->P;
        if (P == 1) {
            if (h->n <= 4) PF_LAUNCH_COUNT(4, 1);
            else if (h->n <= 8) PF_LAUNCH_COUNT(8, 1);
            else PF_LAUNCH_COUNT(PF_NMAX, 1);
        } else if (P == 2) {
            if (h->n <= 4) PF_LAUNCH_COUNT(4, 2);
            else if (h->n <= 8) PF_LAUNCH_COUNT(8, 2);
            else PF_LAUNCH_COUNT(PF_NMAX, 2);
        } else {
            if (h->n <= 8) PF_LAUNCH_COUNT(8, PF_PMAX);
            else PF_LAUNCH_COUNT(PF_NMAX, PF_PMAX);
        }
#undef PF_LAUNCH_COUNT
        h->fin_pending = true;
    }
    return check_launch("k_count");
}

// ancestor-ledger maintenance of this step (no-op unless the step resampled); closes the step on the counting stream
static int launch_ledger(pf_handle* h, long long s) {
    (void)s;
    if (h->no_count) return 0;
    if (h->ev_dec) hipStreamWaitEvent(h->cstream, h->ev_dec, 0);
    hipLaunchKernelGGL(k_ledger, dim3(h->nblocks + PF_LEDGER_BLOCKS), dim3(PF_BS), 0, h->cstream, h->A, h->nblocks);
    h->ev_cnt = next_sync_event(h);
    hipEventRecord(h->ev_cnt, h->cstream);
    return check_launch("k_ledger");
}

static int launch_resample(pf_handle* h, long long s) {
    const bool t = timing_on(h, s);
    {
        Timed tm(h, 3, t);
        hipLaunchKernelGGL(k_resample, dim3(h->nblocks), dim3(PF_BS), 0, h->stream, h->A, s, h->nblocks);
    }
    return check_launch("k_resample");
}

static double seg_pos(pf_handle* h, long long s) {
    return std::min(h->h_seg_start[s] + h->h_seg_len[s], h->h_L);
}

// Single steps.  update and count of one segment share the window set: pf_update_segment evaluates the
// window rule for segment s (the bookkeeping workgroup of k_decide needs it), pf_count launches the sums.
int pf_update_segment(pf_handle* h, int64_t s) {
    HIPCHK(hipSetDevice(h->device));
    if (s < 0 || s >= h->n_segs) { g_err = "segment index out of range"; return -1; }
    h->step_windows = host_windows(h, seg_pos(h, s), false);
    h->A.sp = (int)(s & 1);
    if (launch_extend(h, s)) return -1;
    return launch_decide(h, s, 0, h->step_windows);
}
int pf_count(pf_handle* h, int64_t s, int end_data) {
    HIPCHK(hipSetDevice(h->device));
    (void)end_data;
    return launch_count(h, s, h->step_windows);
}
int pf_resample(pf_handle* h, int64_t s) {
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_resample(h, s);
    if (!rc) rc = launch_ledger(h, s);
    h->seg_done = std::max<long long>(h->seg_done, s + 1);
    return rc;
}

// keep the timing-event pool bounded without stalling the queue: only harvest finished spans
static void trim_spans(pf_handle* h) {
    if (h->spans.empty() || hipEventQuery(h->spans.front().b) != hipSuccess) return;
    size_t done = 0;
    while (done < h->spans.size() && hipEventQuery(h->spans[done].b) == hipSuccess) ++done;
    std::vector<pf_handle::Span> rest(h->spans.begin() + done, h->spans.end());
    h->spans.resize(done);
    harvest_spans(h);
    h->spans = rest;
}

// The single-stream pipeline of the register-tree kernels.  Per row two launches and nothing else:
//   k_row(s)           extend over row s (completing row s-1 while loading)  ||  lagged counts of row s-1
//   k_decide_ledger(s) normalisation / ESS / offspring table of row s        ||  ancestor-ledger upkeep of row s-1
// Stream order provides every dependency; the two halves of each launch touch disjoint (immutable or parity
// double-buffered) data.  The last row of the call is flushed with the stand-alone kernels so that the state is
// whole when the call returns.
template <int NM, bool BIASED>
static void launch_row(pf_handle* h, long long s, int fuse, int count_first, const Windows& Wprev) {
    const size_t smem_reg = (size_t)(2 * PF_EPAD + h->E + 2 * PF_BIAS_MAX + 3) * 8;
    const int nb = h->nblocks;
    const int ncount = count_first < h->E ? nb * (h->E - count_first) : 0;
    if (h->A.rec_trees)
        hipLaunchKernelGGL((k_row<NM, BIASED, false, true>), dim3(nb + ncount), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse, nb, count_first, Wprev);
    else if (h->n == NM)
        hipLaunchKernelGGL((k_row<NM, BIASED, true>), dim3(nb + ncount), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse, nb, count_first, Wprev);
    else
        hipLaunchKernelGGL((k_row<NM, BIASED, false>), dim3(nb + ncount), dim3(PF_BS), smem_reg, h->stream, h->A, s, fuse, nb, count_first, Wprev);
}

static int run_single_stream(pf_handle* h, long long s_begin, long long s_end) {
    const bool biased = h->A.n_bias > 0 || h->A.g_K > 0;
    bool pending = false;             // counts + ledger of the previous row still to be launched
    Windows Wprev = no_windows(h);
    // whatever the two-stream kernels of an earlier call left on the counting stream must be done first
    if (h->ev_cnt) { hipStreamWaitEvent(h->stream, h->ev_cnt, 0); h->ev_cnt = nullptr; }
    for (long long s = s_begin; s < s_end; ++s) {
        h->step_windows = host_windows(h, seg_pos(h, s), false);
        h->A.sp = (int)(s & 1);
        const bool t = timing_on(h, s);
        {
            Timed tm(h, 0, t);
            const int cf = pending ? Wprev.first : h->E;
            const int fuse = s > s_begin ? 1 : 0;
            if (h->n <= 4 && biased) launch_row<4, true>(h, s, fuse, cf, Wprev);
            else if (h->n <= 4) launch_row<4, false>(h, s, fuse, cf, Wprev);
            else if (biased) launch_row<8, true>(h, s, fuse, cf, Wprev);
            else launch_row<8, false>(h, s, fuse, cf, Wprev);
            if (pending && Wprev.first < h->E) h->fin_pending = true;
        }
        if (check_launch("k_row")) return -1;
        {
            Timed tm(h, 1, t);
            // decide and ledger workgroups each hold ~120 KB of LDS, one per CU: keep the launch within one wave of 256 CUs
            const int lroom = 256 - (h->nblocks + 1) - h->nblocks;
            const int lnbt = pending ? h->nblocks + std::max(16, std::min(PF_LEDGER_BLOCKS, lroom)) : 0;
            hipLaunchKernelGGL(k_decide_ledger, dim3(h->nblocks + 1 + lnbt), dim3(PF_LEDGER_MAXT), 0, h->stream, h->A, s, 0, h->step_windows,
                               h->nblocks, lnbt);
        }
        if (check_launch("k_decide")) return -1;
        pending = true;
        Wprev = h->step_windows;
        h->seg_done = s + 1;
        const bool last = (s + 1 == s_end) || (h->h_seg_start[s] + h->h_seg_len[s] >= h->h_L);
        if (last) {
            // flush: complete the row, then its counts and ledger with the stand-alone kernels (same stream)
            if (launch_resample(h, s)) return -1;
            if (Wprev.first < h->E) {
                const dim3 grid(h->nblocks, h->E - Wprev.first), blk(PF_BS);
                if (h->n <= 4) hipLaunchKernelGGL((k_count<4, 1>), grid, blk, 0, h->stream, h->A, Wprev.first, Wprev);
                else hipLaunchKernelGGL((k_count<8, 1>), grid, blk, 0, h->stream, h->A, Wprev.first, Wprev);
                h->k_launches[2] += 1;
                h->fin_pending = true;
            }
            hipLaunchKernelGGL(k_ledger, dim3(h->nblocks + PF_LEDGER_BLOCKS), dim3(PF_BS), 0, h->stream, h->A, h->nblocks);
            if (check_launch("k_count/k_ledger")) return -1;
            break;
        }
        if ((s & 1023) == 1023) trim_spans(h);
    }
    return 0;
}

// The single-launch pipeline (k_pipe): per row ONE launch on one stream, see the kernel's header.  Rows [s_begin, s_end)
// are followed by two flush launches (completion of the last row with the bookkeeping / ledger / counts still owed)
// after which the handle's state is in the form the general kernels expect.
template <int NM, bool BIASED>
static void launch_pipe(pf_handle* h, long long s, const PipeLaunch& PL, int ncount_wg, const Windows& Wb) {
    const dim3 grid((unsigned)(PL.nb + 1 + PL.nL + ncount_wg)), blk(PF_BS);
    if (h->A.rec_trees)
        hipLaunchKernelGGL((k_pipe<NM, BIASED, false, true>), grid, blk, h->smem_pipe, h->stream, h->A, s, PL, Wb);
    else if (h->n == NM)
        hipLaunchKernelGGL((k_pipe<NM, BIASED, true, false>), grid, blk, h->smem_pipe, h->stream, h->A, s, PL, Wb);
    else
        hipLaunchKernelGGL((k_pipe<NM, BIASED, false, false>), grid, blk, h->smem_pipe, h->stream, h->A, s, PL, Wb);
}

static int run_pipeline(pf_handle* h, long long s_begin, long long s_end) {
    if (s_begin >= s_end) return 0;
    const bool biased = h->A.n_bias > 0 || h->A.g_K > 0;
    const int nb = h->nblocks, E = h->E;
    // whatever the two-stream kernels of an earlier call left on the counting stream must be done first
    if (h->ev_cnt) { hipStreamWaitEvent(h->stream, h->ev_cnt, 0); h->ev_cnt = nullptr; }
    hipLaunchKernelGGL(k_pipe_seed, dim3(1), dim3(1), 0, h->stream, h->A, (int)((s_begin + PF_RING - 1) & (PF_RING - 1)));
    Windows W1 = no_windows(h), W2 = no_windows(h);      // windows of rows s-1 and s-2
    long long last = s_begin - 1;                        // last row extended so far
    const int nL_full = nb + h->ledger_wgs;
    auto dispatch = [&](long long s, const PipeLaunch& PL, int ncount_wg, const Windows& Wb) {
        if (h->n <= 4 && biased) launch_pipe<4, true>(h, s, PL, ncount_wg, Wb);
        else if (h->n <= 4) launch_pipe<4, false>(h, s, PL, ncount_wg, Wb);
        else if (biased) launch_pipe<8, true>(h, s, PL, ncount_wg, Wb);
        else launch_pipe<8, false>(h, s, PL, ncount_wg, Wb);
    };
    // one launch: extend row s (or only complete row s-1 / nothing), bookkeeping of row s-1, ledger + counts of row s-2
    auto launch = [&](long long s, bool extend, bool have_b, bool have_lc, int set_cur) -> int {
        PipeLaunch PL;
        memset(&PL, 0, sizeof(PL));
        PL.nb = nb; PL.nblk = nb;
        PL.row.extend = extend ? 1 : 0;
        PL.row.complete = (s > s_begin && s - 1 <= last && (extend || have_b)) ? 1 : 0;
        if (!extend && !have_b) PL.row.complete = 0;
        PL.row.slot_prev = PL.row.complete ? (int)((s - 1) & (PF_RING - 1)) : -1;
        PL.row.slot_out = (int)(s & (PF_RING - 1));
        PL.row.pos_prev = (PL.row.complete && s > s_begin) ? seg_pos(h, s - 1) : 0.0;   // row s - 1 of the second flush step may lie past the table
        PL.b_slot = have_b ? (int)((s - 1) & (PF_RING - 1)) : -1;
        PL.b_row = s - 1;
        PL.b_pos = have_b ? seg_pos(h, s - 1) : 0.0;
        PL.b_set_cur = set_cur;
        PL.lc_slot = (have_lc && !h->no_count) ? (int)((s - 2) & (PF_RING - 1)) : -1;
        PL.live_slot = (int)((s - 1) & (PF_RING - 1));
        PL.nL = PL.lc_slot >= 0 ? nL_full : 0;
        PL.ncw = h->ncw;
        PL.nT = 0; PL.row.draws = 0;               // k_pipe keeps no draw table
        const int ncount_wg = (PL.lc_slot >= 0 && W2.first < E) ? h->cw_off[E - W2.first] : 0;
        if (ncount_wg > 0) h->fin_pending = true;
        const bool t = extend && timing_on(h, s);
        {
            Timed tm(h, 0, t);
            if (!extend) h->k_launches[0] -= 1;          // flush launches are not rows
            dispatch(s, PL, ncount_wg, W1);
        }
        return check_launch("k_pipe");
    };
    long long s = s_begin;
    for (; s < s_end; ++s) {
        if (launch(s, true, s > s_begin, s > s_begin + 1, -1)) return -1;
        last = s;
        W2 = W1;
        W1 = host_windows(h, seg_pos(h, s), false);
        h->step_windows = W1;
        h->seg_done = s + 1;
        if ((s & 1023) == 1023) trim_spans(h);
        if (h->h_seg_start[s] + h->h_seg_len[s] >= h->h_L) { ++s; break; }      // smcsmc.cpp:353-356
    }
    // flush 1: complete row `last` into the next ring slot (the general kernels continue from there), its bookkeeping,
    // ledger + counts of the row before it; flush 2: ledger + counts of row `last`
    if (launch(last + 1, false, true, last - 1 >= s_begin, (int)((last + 1) & (PF_RING - 1)))) return -1;
    W2 = W1;
    if (launch(last + 2, false, false, true, -1)) return -1;
    return 0;
}

// Rows [s_begin, s_end) of several chunks (handles on one device, same shape) in lockstep, one k_sweep launch per step on
// the leader's stream.  Every chunk is bit-identical to its own pf_run (tests/test_gpu_sweep.py): a chunk never reads
// another chunk's memory, and the launch geometry a chunk sees is the one k_pipe gives it.
template <int NM, bool BIASED>
static void launch_sweep(pf_handle* h, const dim3& grid, long long t) {
    const dim3 blk(PF_BS);
    const size_t lds = h->smem_pipe;
    if constexpr (NM == 4 && !BIASED) {
        const bool traced = h->d_trace && t >= h->trace_t0 && t < h->trace_t0 + h->trace_n;
        if (h->h_sweep[0].workers > 0) {
            if (h->n == NM) { if (traced) hipLaunchKernelGGL((k_sweep4q<true, true>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks); else hipLaunchKernelGGL((k_sweep4q<true, false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks); }
            else { if (traced) hipLaunchKernelGGL((k_sweep4q<false, true>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks); else hipLaunchKernelGGL((k_sweep4q<false, false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks); }
            return;
        }
        if (h->d_trace && !h->A.rec_trees && t >= h->trace_t0 && t < h->trace_t0 + h->trace_n) {
            if (h->n == NM) hipLaunchKernelGGL((k_sweep4t<true>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
            else hipLaunchKernelGGL((k_sweep4t<false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
            return;
        }
        if (h->A.rec_trees) hipLaunchKernelGGL((k_sweep4<false, true>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
        else if (h->n == NM) hipLaunchKernelGGL((k_sweep4<true, false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
        else hipLaunchKernelGGL((k_sweep4<false, false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
    } else {
        if (h->A.rec_trees) hipLaunchKernelGGL((k_sweep<NM, BIASED, false, true>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
        else if (h->n == NM) hipLaunchKernelGGL((k_sweep<NM, BIASED, true, false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
        else hipLaunchKernelGGL((k_sweep<NM, BIASED, false, false>), grid, blk, lds, h->stream, h->d_sweep, t, h->nblocks);
    }
}

static bool sweep_compatible(const pf_handle* a, const pf_handle* b) {
    const bool ba = a->A.n_bias > 0 || a->A.g_K > 0, bb = b->A.n_bias > 0 || b->A.g_K > 0;
    return a->device == b->device && a->Np == b->Np && a->n == b->n && a->E == b->E && a->P == b->P && ba == bb &&
           a->A.rec_trees == b->A.rec_trees && a->ncw == b->ncw && a->workers == b->workers && a->ledger_wgs == b->ledger_wgs && a->split_many == b->split_many && a->cw_off == b->cw_off && a->smem_pipe == b->smem_pipe && a->no_count == b->no_count &&
           (a->A.dt_tab != nullptr) == (b->A.dt_tab != nullptr);
}

// the per-chunk table of a k_sweep call in the leader's device buffer; returns the number of steps (0: nothing to do)
static long long sweep_table(pf_handle* const* hs, int nh, long long s_begin, long long s_end, int nL_full, bool* failed) {
    pf_handle* h = hs[0];
    const int E = h->E;
    *failed = true;
    if (h->d_sweep_cap < nh) {
        if (h->d_sweep) { if (hipStreamSynchronize(h->stream) != hipSuccess || hipFree(h->d_sweep) != hipSuccess) return 0; }
        if (hipMalloc((void**)&h->d_sweep, sizeof(SweepChunk) * (size_t)nh) != hipSuccess) { g_err = "hipMalloc of the chunk table failed"; return 0; }
        h->d_sweep_cap = nh;
    }
    // the table of the previous call may still be read by its launches: a fresh host copy per call, uploaded in stream order
    if (hipStreamSynchronize(h->stream) != hipSuccess) { g_err = "hipStreamSynchronize failed"; return 0; }
    h->h_sweep.assign((size_t)nh, SweepChunk());
    long long steps = 0;
    for (int k = 0; k < nh; ++k) {
        pf_handle* g = hs[k];
        SweepChunk& ch = h->h_sweep[k];
        memset(&ch, 0, sizeof(ch));
        ch.A = g->A;
        ch.s_begin = s_begin;
        long long e = std::min<long long>(s_end, g->n_segs), last = s_begin - 1;
        for (long long s = s_begin; s < e; ++s) { last = s; if (g->h_seg_start[s] + g->h_seg_len[s] >= g->h_L) break; }   // smcsmc.cpp:353-356
        ch.s_last = last;
        for (int q = 0; q < E; ++q) ch.counted_to[q] = g->h_counted_to[q];
        ch.no_count = g->no_count ? 1 : 0;
        ch.nL_full = nL_full;
        ch.ncw = g->ncw;
        ch.nblk = g->nblocks;
        ch.nT = (g->A.dt_tab && g->P == 1) ? g->nblocks : 0;
        ch.split = (g->P == 1 && g->split_roles && !g->A.rec_trees) ? 1 : 0;
        ch.workers = g->workers;
        ch.handoff = h->sweep_handoff ? 1 : 0;
        ch.xt_wgs = h->sweep_handoff ? g->nblocks + (ch.nT > 0 ? 1 + ch.nT : 0) : 0;
        if (h->sweep_handoff) ch.split = 1;
        if (h->sweep_split2) ch.split = 2;
        ch.trace = h->d_trace; ch.trace_t0 = h->trace_t0; ch.trace_n = h->d_trace ? h->trace_n : 0; ch.trace_stride = h->trace_stride;
        if (last >= s_begin) steps = std::max(steps, last - s_begin + 3);
    }
    *failed = false;
    if (steps == 0) return 0;
    if (hipMemcpyAsync(h->d_sweep, h->h_sweep.data(), sizeof(SweepChunk) * (size_t)nh, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
        g_err = "upload of the chunk table failed"; *failed = true; return 0;
    }
    hipLaunchKernelGGL(k_sweep_seed, dim3(nh), dim3(PF_BS), 0, h->stream, h->d_sweep);
    return steps;
}

static int run_sweep(pf_handle* const* hs, int nh, long long s_begin, long long s_end) {
    pf_handle* h = hs[0];
    if (s_begin >= s_end) return 0;
    const int nb = h->nblocks, E = h->E;
    const bool biased = h->A.n_bias > 0 || h->A.g_K > 0;
    const int nL_full = nb + h->ledger_wgs;
    // the other handles' streams (anything they still have in flight) come first, then the table
    for (int k = 0; k < nh; ++k) {
        pf_handle* g = hs[k];
        if (g->ev_cnt) { hipStreamWaitEvent(h->stream, g->ev_cnt, 0); g->ev_cnt = nullptr; }
        if (k > 0) {
            hipEvent_t ev = next_sync_event(g);
            hipEventRecord(ev, g->stream);
            hipStreamWaitEvent(h->stream, ev, 0);
        }
    }
    if (h->trace_n > 0 && !h->d_trace) {
        // pf_set_wg_trace: room for the largest grid a step of these chunks can have
        const int nT = (h->A.dt_tab && h->P == 1) ? nb : 0;
        h->trace_stride = nh * (nb + 1 + nT + nL_full + h->cw_off[E]);
        h->trace_words = (size_t)h->trace_n * (size_t)h->trace_stride * 4;
        if (hipMalloc((void**)&h->d_trace, h->trace_words * 8) != hipSuccess) { h->d_trace = nullptr; g_err = "hipMalloc of the workgroup trace failed"; return -1; }
        hipMemsetAsync(h->d_trace, 0, h->trace_words * 8, h->stream);
    }
    bool failed = false;
    const long long steps = sweep_table(hs, nh, s_begin, s_end, nL_full, &failed);
    if (failed) return -1;
    if (steps == 0) return 0;
    // the count workgroups of a step belong to row s - 2, whose windows the host knows as well as the device does: the grid
    // ends with the last epoch column any chunk needs (epochs before a chunk's first moving one are not launched at all)
    std::vector<Windows> W1((size_t)nh), W2((size_t)nh);
    for (int k = 0; k < nh; ++k) { W1[k] = no_windows(hs[k]); W2[k] = W1[k]; }
    for (long long t = 0; t < steps; ++t) {
        const long long s = s_begin + t;
        int columns = 0;
        for (int k = 0; k < nh; ++k)
            if (s >= s_begin + 2 && s - 2 <= h->h_sweep[k].s_last && !hs[k]->no_count) columns = std::max(columns, E - W2[k].first);
        const int ncount = h->cw_off[columns];
        bool any_lc = false;
        for (int k = 0; k < nh; ++k) any_lc = any_lc || (s >= s_begin + 2 && s - 2 <= h->h_sweep[k].s_last && !hs[k]->no_count);
        const unsigned per_chunk = (unsigned)(nb + 1 + h->h_sweep[0].nT + (h->h_sweep[0].workers > 0 ? (any_lc ? h->h_sweep[0].workers : 0) : nL_full + ncount));
        const dim3 grid(per_chunk, (unsigned)nh);          // (pf_bx() / pf_chunk(), pf_device.h)
        const bool tm_on = timing_on(h, s);
        {
            Timed tm(h, 0, tm_on);
            if (h->n <= 4 && biased) launch_sweep<4, true>(h, grid, t);
            else if (h->n <= 4) launch_sweep<4, false>(h, grid, t);
            else if (biased) launch_sweep<8, true>(h, grid, t);
            else launch_sweep<8, false>(h, grid, t);
        }
        if (check_launch("k_sweep")) return -1;
        if ((t & 1023) == 1023) trim_spans(h);
        for (int k = 0; k < nh; ++k) {
            pf_handle* g = hs[k];
            W2[k] = W1[k];
            if (s <= h->h_sweep[k].s_last) {
                W1[k] = host_windows(g, seg_pos(g, s), false);
                g->step_windows = W1[k];
                if (W1[k].first < E && !g->no_count) g->fin_pending = true;
            } else {
                W1[k] = no_windows(g);
            }
        }
    }
    h->k_launches[0] -= 2;                                      // flush steps are not rows
    for (int k = 0; k < nh; ++k) {
        pf_handle* g = hs[k];
        const long long last = h->h_sweep[k].s_last;
        if (last >= s_begin) g->seg_done = last + 1;
        if (k > 0) {                                            // the chunk's own stream continues after the sweep
            hipEvent_t ev = next_sync_event(h);
            hipEventRecord(ev, h->stream);
            hipStreamWaitEvent(g->stream, ev, 0);
        }
    }
    return 0;
}

// One or several chunks in lockstep as TWO launches per step (the default for one population, at most four haplotypes, no focused sampling;
// PF_DEBUG_ONE_LAUNCH = run_sweep): the extend, bookkeeping and draw roles of all chunks
// (k_sweep4, one launch after the other on the leader's stream: the chain of dependent loads that is the critical path of a step) and
// their ledger and count roles (k_sweep_blc, on the counting stream, behind the extend launch of the step before by its completion
// signal and paced by the sixteen-slot ring as in run_sweep_mp).  The second launch needs no dynamic LDS -- that is the bookkeeping
// role's -- and fewer registers than the extend role: four of its workgroups share a compute unit where the single launch has room for
// three, and its tail no longer holds up the next row's extend role.  Same bits as run_sweep.
static int run_sweep_split(pf_handle* const* hs, int nh, long long s_begin, long long s_end) {
    pf_handle* h = hs[0];
    if (s_begin >= s_end) return 0;
    const int nb = h->nblocks, E = h->E;
    const int nL_full = nb + h->ledger_wgs;
    for (int k = 0; k < nh; ++k) {
        pf_handle* g = hs[k];
        if (g->ev_cnt) { hipStreamWaitEvent(h->stream, g->ev_cnt, 0); g->ev_cnt = nullptr; }
        if (k > 0) {
            hipEvent_t ev = next_sync_event(g);
            hipEventRecord(ev, g->stream);
            hipStreamWaitEvent(h->stream, ev, 0);
        }
    }
    if (h->ev_x.empty()) {
        std::vector<hipEvent_t> ev(32, nullptr);
        bool ok = true;
        for (auto& e : ev) if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { e = nullptr; ok = false; }
        if (!ok) { for (auto e : ev) if (e) hipEventDestroy(e); g_err = "hipEventCreate failed"; return -1; }
        h->ev_x.assign(ev.begin(), ev.begin() + 16); h->ev_blc.assign(ev.begin() + 16, ev.end());
    }
    bool failed = false;
    h->sweep_split2 = true;
    const long long steps = sweep_table(hs, nh, s_begin, s_end, nL_full, &failed);
    h->sweep_split2 = false;
    if (failed) return -1;
    if (steps == 0) return 0;
    hipEvent_t seeded = next_sync_event(h);
    hipEventRecord(seeded, h->stream);
    const int nT = h->h_sweep[0].nT;
    const dim3 gx((unsigned)(nb + 1 + nT), (unsigned)nh), blk(PF_BS);
    std::vector<Windows> W1((size_t)nh), W2((size_t)nh);
    for (int k = 0; k < nh; ++k) { W1[k] = no_windows(hs[k]); W2[k] = W1[k]; }
    // (at most four: the second launch of step t - 7 must be enqueued when step t waits for it.  Several chunks: eight 20 Mb chunks 1.205e5 ->
    // 1.248e5 segments/s with two, 1.272e5 with four; one chunk is the same with any)
    const long long batch = std::max(1, std::min(4, h->split_batch > 0 ? h->split_batch : (nh > 1 ? 4 : 1)));
    std::vector<unsigned> pending_grid((size_t)batch, 1u);
    hipStreamWaitEvent(h->cstream, seeded, 0);
    for (long long t = 0; t < steps; ++t) {
        const long long s = s_begin + t;
        static_assert(PF_RING == 16, "the wait schedule below is written for sixteen ring slots");
        if (t >= 8 && (t & 7) == 0) hipStreamWaitEvent(h->stream, h->ev_blc[(size_t)((t - 7) & 15)], 0);      // ring slot reuse, as in run_sweep_mp
        // The second launches follow in batches of `batch` steps: only the last extend launch of a batch carries a completion signal, and the
        // batch's second launches wait for that one (each needs the extend launch of the step before it: complete by then).
        const bool batch_end = ((t + 1) % batch) == 0 || t == steps - 1;
        {
            Timed tm(h, 0, timing_on(h, s));
            hipEvent_t xdone = h->ev_x[(size_t)(t & 15)];
            if (batch_end) {
                if (h->n == 4) hipExtLaunchKernelGGL((k_sweep4<true, false>), gx, blk, h->smem_pipe, h->stream, nullptr, xdone, 0, h->d_sweep, t, nb);
                else hipExtLaunchKernelGGL((k_sweep4<false, false>), gx, blk, h->smem_pipe, h->stream, nullptr, xdone, 0, h->d_sweep, t, nb);
            } else {
                if (h->n == 4) hipLaunchKernelGGL((k_sweep4<true, false>), gx, blk, h->smem_pipe, h->stream, h->d_sweep, t, nb);
                else hipLaunchKernelGGL((k_sweep4<false, false>), gx, blk, h->smem_pipe, h->stream, h->d_sweep, t, nb);
            }
        }
        if (check_launch("k_sweep4 (extend, bookkeeping and draw roles)")) return -1;
        int columns = 0;
        for (int k = 0; k < nh; ++k)
            if (s >= s_begin + 2 && s - 2 <= h->h_sweep[k].s_last && !hs[k]->no_count) columns = std::max(columns, E - W2[k].first);
        bool any_lc = false;
        for (int k = 0; k < nh; ++k) any_lc = any_lc || (s >= s_begin + 2 && s - 2 <= h->h_sweep[k].s_last && !hs[k]->no_count);
        const int W = h->h_sweep[0].workers;
        pending_grid[(size_t)(t % batch)] = (unsigned)(1 + (W > 0 ? (any_lc ? W : 0) : nL_full + h->cw_off[columns]));
        if (batch_end) {
            hipStreamWaitEvent(h->cstream, h->ev_x[(size_t)(t & 15)], 0);
            for (long long u = t - (t % batch); u <= t; ++u) {
                const dim3 grid(pending_grid[(size_t)(u % batch)], (unsigned)nh);
                hipEvent_t bdone = h->ev_blc[(size_t)(u & 15)];
                if (W > 0) {
                    if (h->n == 4) hipExtLaunchKernelGGL((k_sweep_blc4q<true>), grid, blk, 0, h->cstream, nullptr, bdone, 0, h->d_sweep, u);
                    else hipExtLaunchKernelGGL((k_sweep_blc4q<false>), grid, blk, 0, h->cstream, nullptr, bdone, 0, h->d_sweep, u);
                } else {
                    if (h->n == 4) hipExtLaunchKernelGGL((k_sweep_blc4<true>), grid, blk, 0, h->cstream, nullptr, bdone, 0, h->d_sweep, u);
                    else hipExtLaunchKernelGGL((k_sweep_blc4<false>), grid, blk, 0, h->cstream, nullptr, bdone, 0, h->d_sweep, u);
                }
            }
            if (check_launch("k_sweep_blc (ledger and count roles)")) return -1;
        }
        if ((t & 1023) == 1023) trim_spans(h);
        for (int k = 0; k < nh; ++k) {
            pf_handle* g = hs[k];
            W2[k] = W1[k];
            if (s <= h->h_sweep[k].s_last) {
                W1[k] = host_windows(g, seg_pos(g, s), false);
                g->step_windows = W1[k];
                if (W1[k].first < E && !g->no_count) g->fin_pending = true;
            } else {
                W1[k] = no_windows(g);
            }
        }
    }
    h->k_launches[0] -= 2;                                      // flush steps are not rows
    // what follows on any chunk's stream waits for the last launch of the other roles
    hipStreamWaitEvent(h->stream, h->ev_blc[(size_t)((steps - 1) & 15)], 0);
    for (int k = 0; k < nh; ++k) {
        pf_handle* g = hs[k];
        const long long last = h->h_sweep[k].s_last;
        if (last >= s_begin) g->seg_done = last + 1;
        if (k > 0) {
            hipEvent_t ev = next_sync_event(h);
            hipEventRecord(ev, h->stream);
            hipStreamWaitEvent(g->stream, ev, 0);
        }
    }
    return 0;
}

// Structured models (register-tree kernel) on the row pipeline.  Per step two launches: the extend role (k_sweep_xmp, with the
// decision on the previous row in its prologue) on the filter stream, the bookkeeping / ledger / count roles (k_sweep_blc) on
// the counting stream -- separate launches because the extend workgroups' LDS (their trees' migration events) would be
// allocated to every count workgroup too.  Step t's second launch needs the extend launch of step t - 1 (partials, offspring
// table, records), the extend launch of step t must not overwrite ring slot t & (PF_RING - 1) before the counts of step t - 2 are done:
// one wait each way per step, on the kernels' own completion signals; the extend role does not read anything the other
// launch writes (it keeps its own note of n_resample / generation, Ctrl::xr).  k_decide, its boundary and the wait of the
// next row on the previous row's ledger upkeep are gone from the critical stream.
static int run_sweep_mp(pf_handle* h, long long s_begin, long long s_end) {
    if (s_begin >= s_end) return 0;
    const int nb = h->nblocks, E = h->E;
    const bool biased = h->A.n_bias > 0 || h->A.g_K > 0;
    const int nL_full = nb + h->ledger_wgs;
    if (h->ev_cnt) { hipStreamWaitEvent(h->stream, h->ev_cnt, 0); h->ev_cnt = nullptr; }
    if (h->ev_x.empty()) {
        // all thirty-two or none: a vector left half filled would pass for complete on the next call, and launches with null
        // completion events lose the ordering between the two streams without a word
        std::vector<hipEvent_t> ev(32, nullptr);
        bool ok = true;
        for (auto& e : ev) if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { e = nullptr; ok = false; }
        if (!ok) {
            for (auto e : ev) if (e) hipEventDestroy(e);
            g_err = "hipEventCreate failed";
            return -1;
        }
        h->ev_x.assign(ev.begin(), ev.begin() + 16); h->ev_blc.assign(ev.begin() + 16, ev.end());
    }
    pf_handle* one[1] = {h};
    bool failed = false;
    const long long steps = sweep_table(one, 1, s_begin, s_end, nL_full, &failed);
    if (failed) return -1;
    if (steps == 0) return 0;
    // the counting stream starts behind the table and the seed
    hipEvent_t seeded = next_sync_event(h);
    hipEventRecord(seeded, h->stream);
    Windows W1 = no_windows(h), W2 = W1;
    const long long last = h->h_sweep[0].s_last;
    for (long long t = 0; t < steps; ++t) {
        const long long s = s_begin + t;
        // ring slot reuse: the extend launch of step t overwrites the slot of row t - PF_RING, which the counts read in step
        // t - PF_RING + 2.  One wait every eight steps, for the second launch of seven steps ago, covers the eight steps that
        // follow (t + 7 - 14 <= t - 7); between two waits the extend launches are dispatched back to back.
        static_assert(PF_RING == 16, "the wait schedule below is written for sixteen ring slots");
        if (t >= 8 && (t & 7) == 0) hipStreamWaitEvent(h->stream, h->ev_blc[(size_t)((t - 7) & 15)], 0);
        {
            Timed tm(h, 0, timing_on(h, s));
            if (h->P > 1) pf_mp_launch_sweep_x(h->A, h->d_sweep, t, h->smem_sweep_x, h->stream, h->ev_x[(size_t)(t & 15)]);
            else {
                // one population (PF_DEBUG_SPLIT_ROLES): k_sweep with a grid of the extend workgroups only
                const dim3 gx((unsigned)(nb + (h->h_sweep[0].nT > 0 ? 1 + h->h_sweep[0].nT : 0)), 1u), bx(PF_BS);
                hipEvent_t xdone = h->ev_x[(size_t)(t & 15)];
#define PF_LAUNCH_X(NMV, BV, EV) hipExtLaunchKernelGGL((k_sweep<NMV, BV, EV, false>), gx, bx, h->smem_pipe, h->stream, nullptr, xdone, 0, h->d_sweep, t, nb)
                if (h->n <= 4) { if (biased) { if (h->n == 4) PF_LAUNCH_X(4, true, true); else PF_LAUNCH_X(4, true, false); } else { if (h->n == 4) PF_LAUNCH_X(4, false, true); else PF_LAUNCH_X(4, false, false); } }
                else { if (biased) { if (h->n == 8) PF_LAUNCH_X(8, true, true); else PF_LAUNCH_X(8, true, false); } else { if (h->n == 8) PF_LAUNCH_X(8, false, true); else PF_LAUNCH_X(8, false, false); } }
#undef PF_LAUNCH_X
            }
        }
        if (check_launch("k_sweep (extend role)")) return -1;
        hipStreamWaitEvent(h->cstream, t >= 1 ? h->ev_x[(size_t)((t - 1) & 15)] : seeded, 0);
        const int columns = (s >= s_begin + 2 && s - 2 <= last && !h->no_count) ? E - W2.first : 0;
        const int ncount = h->cw_off[columns];
        const dim3 grid((unsigned)(1 + (h->h_sweep[0].workers > 0 ? ((s >= s_begin + 2 && s - 2 <= last && !h->no_count) ? h->h_sweep[0].workers : 0) : nL_full + ncount)), 1u), blk(PF_BS);
        hipEvent_t done = h->ev_blc[(size_t)(t & 15)];
#define PF_LAUNCH_BLC(NMV, PV, BV) hipExtLaunchKernelGGL((k_sweep_blc<NMV, PV, BV>), grid, blk, h->smem_pipe, h->cstream, nullptr, done, 0, h->d_sweep, t)
        if (h->P == 1) { if (h->n <= 4) { if (biased) PF_LAUNCH_BLC(4, 1, true); else PF_LAUNCH_BLC(4, 1, false); } else { if (biased) PF_LAUNCH_BLC(8, 1, true); else PF_LAUNCH_BLC(8, 1, false); } }
        else if (h->P == 2) { if (biased) PF_LAUNCH_BLC(8, 2, true); else PF_LAUNCH_BLC(8, 2, false); }
        else { if (biased) PF_LAUNCH_BLC(8, PF_PMAX, true); else PF_LAUNCH_BLC(8, PF_PMAX, false); }
#undef PF_LAUNCH_BLC
        if (check_launch("k_sweep_blc")) return -1;
        if ((t & 1023) == 1023) trim_spans(h);
        W2 = W1;
        if (s <= last) {
            W1 = host_windows(h, seg_pos(h, s), false);
            h->step_windows = W1;
            if (W1.first < E && !h->no_count) h->fin_pending = true;
        } else {
            W1 = no_windows(h);
        }
    }
    h->k_launches[0] -= 2;                                      // flush steps are not rows
    h->ev_cnt = h->ev_blc[(size_t)((steps - 1) & 15)];       // what follows on the filter stream waits for the last counts
    if (last >= s_begin) h->seg_done = last + 1;
    return 0;
}

// One population, rows handed over through memory (PF_DEBUG_FLAG_HANDOFF).  A row is two launches, as in the split arrangement of
// run_sweep_mp -- k_sweep with the extend and draw roles, k_sweep_blc with bookkeeping, ledger and counts -- but no launch waits for
// another launch to END: the extend / draw launches alternate between two streams, so that step t + 1 is dispatched while step t
// still runs, and its workgroups wait in the kernel for the arrivals of step t's workgroups (Ctrl::xt_done; sweep_wait_ge); the
// launches of the other roles follow one another on the counting stream and wait the same way for the extend launch they read
// from; ring reuse (the extend launch of step t overwrites what the counts of step t - 14 read) is a wait on Ctrl::blc_step.  What
// a row pays for the hand-off is then a release, an arrival and a poll (3.8 us for 157 wavefronts, pf_probe_handoff) instead of a
// kernel boundary (5.4 us), and the count workgroups no longer share the launch of the extend workgroups.
static int run_sweep_flags(pf_handle* h, long long s_begin, long long s_end) {
    if (s_begin >= s_end) return 0;
    const int nb = h->nblocks, E = h->E;
    const bool biased = h->A.n_bias > 0 || h->A.g_K > 0;
    const int nL_full = nb + h->ledger_wgs;
    if (h->ev_cnt) { hipStreamWaitEvent(h->stream, h->ev_cnt, 0); h->ev_cnt = nullptr; }
    pf_handle* one[1] = {h};
    bool failed = false;
    // (sweep_table uploads the chunk table and runs the seed on h->stream; the hand-off fields are set in the host copy first)
    h->sweep_handoff = true;
    const long long steps = sweep_table(one, 1, s_begin, s_end, nL_full, &failed);
    h->sweep_handoff = false;
    if (failed) return -1;
    if (steps == 0) return 0;
    if (h->ev_x.empty()) {
        std::vector<hipEvent_t> ev(32, nullptr);
        bool ok = true;
        for (auto& e : ev) if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { e = nullptr; ok = false; }
        if (!ok) { for (auto e : ev) if (e) hipEventDestroy(e); g_err = "hipEventCreate failed"; return -1; }
        h->ev_x.assign(ev.begin(), ev.begin() + 16); h->ev_blc.assign(ev.begin() + 16, ev.end());
    }
    hipEvent_t seeded = next_sync_event(h);
    hipEventRecord(seeded, h->stream);
    hipStreamWaitEvent(h->stream2, seeded, 0);
    hipStreamWaitEvent(h->cstream, seeded, 0);
    const int nT = h->h_sweep[0].nT;
    const dim3 gx((unsigned)(nb + (nT > 0 ? 1 + nT : 0)), 1u), bx(PF_BS);
    Windows W1 = no_windows(h), W2 = W1;
    const long long last = h->h_sweep[0].s_last;
    for (long long t = 0; t < steps; ++t) {
        const long long s = s_begin + t;
        hipStream_t xs = (t & 1) ? h->stream2 : h->stream;
        hipEvent_t xdone = h->ev_x[(size_t)(t & 15)];
        {
            Timed tm(h, 0, timing_on(h, s) && !(t & 1));
#define PF_LAUNCH_XF(KERN) hipExtLaunchKernelGGL((KERN), gx, bx, h->smem_pipe, xs, nullptr, xdone, 0, h->d_sweep, t, nb)
            if (h->n == 4) PF_LAUNCH_XF((k_sweep4h<true>)); else PF_LAUNCH_XF((k_sweep4h<false>));
#undef PF_LAUNCH_XF
        }
        if (check_launch("k_sweep (extend role, flag hand-off)")) return -1;
        // the other roles' launch of step t reads what the extend launch of step t - 1 wrote: ordered by that launch's completion
        // signal on the counting stream (a queue-level wait: nothing spins; this stream is not the critical one)
        if (t >= 1) hipStreamWaitEvent(h->cstream, h->ev_x[(size_t)((t - 1) & 15)], 0);
        const int columns = (s >= s_begin + 2 && s - 2 <= last && !h->no_count) ? E - W2.first : 0;
        const int ncount = h->cw_off[columns];
        const dim3 grid((unsigned)(1 + (h->h_sweep[0].workers > 0 ? ((s >= s_begin + 2 && s - 2 <= last && !h->no_count) ? h->h_sweep[0].workers : 0) : nL_full + ncount)), 1u), blk(PF_BS);
#define PF_LAUNCH_BLCF(NMV, BV) hipLaunchKernelGGL((k_sweep_blc<NMV, 1, BV>), grid, blk, h->smem_pipe, h->cstream, h->d_sweep, t)
        if (h->n <= 4) { if (biased) PF_LAUNCH_BLCF(4, true); else PF_LAUNCH_BLCF(4, false); } else { if (biased) PF_LAUNCH_BLCF(8, true); else PF_LAUNCH_BLCF(8, false); }
#undef PF_LAUNCH_BLCF
        if (check_launch("k_sweep_blc (flag hand-off)")) return -1;
        if ((t & 1023) == 1023) trim_spans(h);
        W2 = W1;
        if (s <= last) {
            W1 = host_windows(h, seg_pos(h, s), false);
            h->step_windows = W1;
            if (W1.first < E && !h->no_count) h->fin_pending = true;
        } else {
            W1 = no_windows(h);
        }
    }
    h->k_launches[0] -= 2;                                      // flush steps are not rows
    // what follows on the filter stream waits for all three
    hipEvent_t e2 = next_sync_event(h), ec = next_sync_event(h);
    hipEventRecord(e2, h->stream2); hipEventRecord(ec, h->cstream);
    hipStreamWaitEvent(h->stream, e2, 0); hipStreamWaitEvent(h->stream, ec, 0);
    if (last >= s_begin) h->seg_done = last + 1;
    return 0;
}

// why pf_run_many would refuse these handles (null: it would not)
static const char* run_many_refusal(pf_handle* const* handles, int32_t n_handles) {
    if (n_handles < 1 || !handles || !handles[0]) return "pf_run_many: no handles";
    pf_handle* h = handles[0];
    for (int k = 0; k < n_handles; ++k) {
        pf_handle* g = handles[k];
        if (!g) return "pf_run_many: null handle";
        if (!(extend_can_fuse(g) && g->pipe && !g->two_launch_rows))
            return "pf_run_many: the chunks must run on the single-launch row pipeline (one population, at most 8 haplotypes, no look-ahead)";
        if (!sweep_compatible(h, g)) return "pf_run_many: the chunks must share device, particle count, haplotypes, epochs and options";
        for (int j = 0; j < k; ++j) if (handles[j] == g) return "pf_run_many: a handle appears twice";
    }
    return nullptr;
}

int pf_can_run_many(pf_handle* const* handles, int32_t n_handles) {
    return run_many_refusal(handles, n_handles) == nullptr ? 1 : 0;
}

int pf_run_many(pf_handle* const* handles, int32_t n_handles, int64_t s_begin, int64_t s_end) {
    if (const char* why = run_many_refusal(handles, n_handles)) { g_err = why; return -1; }
    pf_handle* h = handles[0];
    HIPCHK(hipSetDevice(h->device));
    if (s_begin < 0) { g_err = "segment range out of bounds"; return -1; }        // a chunk with fewer rows sits the call out
    if (h->split_many && !h->A.rec_trees) return run_sweep_split(handles, n_handles, s_begin, s_end);
    return run_sweep(handles, n_handles, s_begin, s_end);
}

int pf_run(pf_handle* h, int64_t s_begin, int64_t s_end) {
    HIPCHK(hipSetDevice(h->device));
    if (s_begin < 0 || s_end > h->n_segs) { g_err = "segment range out of bounds"; return -1; }
    if (extend_can_fuse(h)) {
        if (!h->pipe || h->two_launch_rows) return run_single_stream(h, s_begin, s_end);
        if (h->use_k_pipe) return run_pipeline(h, s_begin, s_end);
        if (h->flag_handoff) return run_sweep_flags(h, s_begin, s_end);
        if (h->split_roles && !h->A.rec_trees) return run_sweep_mp(h, s_begin, s_end);
        pf_handle* one[1] = {h};
        if (h->split_many && !h->A.rec_trees) return run_sweep_split(one, 1, s_begin, s_end);
        return run_sweep(one, 1, s_begin, s_end);
    }
    if (h->pipe_mp && h->A.apf == 0 && !h->force_lds && !h->no_fuse) return run_sweep_mp(h, s_begin, s_end);
    // structured models on the register-tree kernel: the next row's extend completes this row while it loads (two
    // launches per row on the main stream instead of three); the last row of the call is completed by k_resample
    const bool mp_fuse = h->P > 1 && pf_mp_can_fuse(h->A, h->force_lds) && h->A.apf == 0 && !h->no_fuse;
    long long owed = -1;              // row decided but not completed yet
    for (long long s = s_begin; s < s_end; ++s) {
        h->step_windows = host_windows(h, seg_pos(h, s), false);
        h->A.sp = (int)(s & 1);
        if (launch_extend(h, s, owed >= 0 ? 1 : 0)) return -1;
        if (launch_decide(h, s, 0, h->step_windows)) return -1;
        if (mp_fuse) owed = s;
        else if (launch_resample(h, s)) return -1;
        if (launch_count(h, s, h->step_windows)) return -1;
        if (launch_ledger(h, s)) return -1;
        h->seg_done = s + 1;
        if (h->h_seg_start[s] + h->h_seg_len[s] >= h->h_L) break;   // smcsmc.cpp:353-356
        if ((s & 1023) == 1023) trim_spans(h);
    }
    if (owed >= 0 && launch_resample(h, owed)) return -1;
    return 0;
}

int pf_finish(pf_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    // smcsmc.cpp:371: normalize_probability once more, then the lag-free flush (373)
    h->A.sp = (int)(h->seg_done & 1);
    if (h->ev_cnt) hipStreamWaitEvent(h->stream, h->ev_cnt, 0);     // k_partials rewrites the parity buffers
    hipLaunchKernelGGL(k_partials, dim3(h->nblocks), dim3(PF_BS), 0, h->stream, h->A);
    h->step_windows = host_windows(h, h->h_L, true);
    if (launch_decide(h, 0, 1, h->step_windows)) return -1;
    if (launch_resample(h, 0)) return -1;    // flag == 0 in mode 1: in-place normalisation
    if (launch_count(h, 0, h->step_windows)) return -1;
    if (launch_ledger(h, 0)) return -1;
    h->finished = true;
    return pf_sync(h);
}

int64_t pf_num_segments_done(pf_handle* h) { return h->seg_done; }

double pf_logl(pf_handle* h) {
    if (pf_sync(h)) return NAN;
    Ctrl c;
    if (hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost) != hipSuccess) return NAN;
    return c.logl;
}

int pf_get_counts(pf_handle* h, double* out, int32_t n) {
    if (pf_sync(h)) return -1;
    const int E = h->E, P = h->P;
    if (n < PF_COUNTS_LEN2(E, P)) { g_err = "count buffer too small"; return -1; }
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    double* tail;
    if (P == 1) {
        HIPCHK(hipMemcpy(out, h->A.totals, (size_t)6 * E * 8, hipMemcpyDeviceToHost));
        tail = out + 6 * E;
    } else {
        // device layout: totals[column][epoch]; columns as AccT<P>.  A kernel compiled for PF_PMAX populations
        // (P == 3) still uses the column numbering of its template parameter.
        const int PT = P == 2 ? 2 : PF_PMAX;
        const int ncol = h->A.ncol;
        std::vector<double> tot((size_t)ncol * E);
        HIPCHK(hipMemcpy(tot.data(), h->A.totals, tot.size() * 8, hipMemcpyDeviceToHost));
        auto col = [&](int k, int e) { return tot[(size_t)k * E + e]; };
        double* o = out;
        for (int stat = 0; stat < 3; ++stat) {                 // coal count, opp, weight [E][P]
            for (int e = 0; e < E; ++e) for (int a = 0; a < P; ++a) o[e * P + a] = col(stat * PT + a, e);
            o += E * P;
        }
        for (int stat = 0; stat < 3; ++stat) {                 // recomb count, opp, weight [E]
            for (int e = 0; e < E; ++e) o[e] = col(3 * PT + stat, e);
            o += E;
        }
        for (int e = 0; e < E; ++e) for (int a = 0; a < P; ++a) for (int b = 0; b < P; ++b)
            o[(e * P + a) * P + b] = col(3 * PT + 3 + a * PT + b, e);
        o += E * P * P;
        for (int e = 0; e < E; ++e) for (int a = 0; a < P; ++a) o[e * P + a] = col(3 * PT + 3 + PT * PT + a, e);
        o += E * P;
        for (int e = 0; e < E; ++e) for (int a = 0; a < P; ++a) o[e * P + a] = col(3 * PT + 3 + PT * PT + PT + a, e);
        o += E * P;
        tail = o;
    }
    tail[0] = c.delayed_opp;
    tail[1] = c.delayed_count;
    tail[2] = (double)c.n_resample;
    tail[3] = c.logl;
    return 0;
}

int pf_get_local_recomb(pf_handle* h, double* opp_diff, double* counts, int64_t nbins) {
    if (pf_sync(h)) return -1;
    if (!h->A.lmap_opp) { g_err = "pf_get_local_recomb: the map was not recorded (pf_params.flags bit 0)"; return -1; }
    const long long nb = h->A.lmap_bins;
    std::vector<double> o(nb), c((size_t)(h->n + 2) * nb);
    HIPCHK(hipMemcpy(o.data(), h->A.lmap_opp, o.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c.data(), h->A.lmap_cnt, c.size() * 8, hipMemcpyDeviceToHost));
    for (long long b = 0; b < nbins; ++b) {
        opp_diff[b] = b < nb ? o[b] : 0.0;
        for (int k = 0; k < h->n + 2; ++k) counts[(size_t)k * nbins + b] = b < nb ? c[(size_t)k * nb + b] : 0.0;
    }
    return 0;
}

int pf_get_migrations(pf_handle* h, int32_t* n_events, double* times, int8_t* branch, int8_t* newpop, int8_t* node_pops,
                      int32_t cap) {
    if (pf_sync(h)) return -1;
    if (h->P < 2) { g_err = "pf_get_migrations: the model has one population"; return -1; }
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    const DState st = state_slot(h->A, c.cur);
    const long long Np = h->Np;
    const int n = h->n;
    std::vector<int> nm(Np);
    HIPCHK(hipMemcpy(nm.data(), st.nm, Np * 4, hipMemcpyDeviceToHost));
    const int mcap = h->A.mcap;
    std::vector<double> mt((size_t)mcap * Np);
    std::vector<int8_t> mb((size_t)mcap * Np), mq((size_t)mcap * Np), pn((size_t)(n - 1) * Np);
    HIPCHK(hipMemcpy(mt.data(), st.Mt, mt.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mb.data(), st.Mb, mb.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mq.data(), st.Mq, mq.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pn.data(), st.Pn, pn.size(), hipMemcpyDeviceToHost));
    for (long long p = 0; p < Np; ++p) {
        if (n_events) n_events[p] = nm[p];
        for (int k = 0; k < cap; ++k) {
            bool ok = k < nm[p] && k < mcap;
            if (times) times[p * cap + k] = ok ? mt[(size_t)k * Np + p] : 0.0;
            if (branch) branch[p * cap + k] = ok ? mb[(size_t)k * Np + p] : 0;
            if (newpop) newpop[p * cap + k] = ok ? (mq[(size_t)k * Np + p] & 3) : 0;      // the upper bits hold the event's epoch
        }
        if (node_pops) for (int r = 0; r < n - 1; ++r) node_pops[p * (n - 1) + r] = pn[(size_t)r * Np + p];
    }
    return 0;
}

int pf_get_trace(pf_handle* h, double* T, double* ess, int32_t* resampled, double* logl, int64_t n) {
    if (pf_sync(h)) return -1;
    long long m = std::min<long long>(n, h->seg_done);
    if (m <= 0) return 0;
    if (T) HIPCHK(hipMemcpy(T, h->A.tr_T, m * 8, hipMemcpyDeviceToHost));
    if (ess) HIPCHK(hipMemcpy(ess, h->A.tr_ess, m * 8, hipMemcpyDeviceToHost));
    if (logl) HIPCHK(hipMemcpy(logl, h->A.tr_logl, m * 8, hipMemcpyDeviceToHost));
    if (resampled) HIPCHK(hipMemcpy(resampled, h->A.tr_flag, m * 4, hipMemcpyDeviceToHost));
    return (int)m;
}

int pf_get_resample_events(pf_handle* h, int32_t* seg_idx, int32_t* parents, int32_t max_events) {
    if (pf_sync(h)) return -1;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    int nev = (int)std::min<long long>(std::min<long long>(c.n_resample, h->max_trace_events), max_events);
    if (nev <= 0) return 0;
    if (seg_idx) HIPCHK(hipMemcpy(seg_idx, h->A.ev_seg, (size_t)nev * 4, hipMemcpyDeviceToHost));
    if (parents) HIPCHK(hipMemcpy(parents, h->A.ev_parents, (size_t)nev * h->Np * 4, hipMemcpyDeviceToHost));
    return nev;
}

int pf_get_particles(pf_handle* h, double* w_post, double* w_pilot, double* heights, int8_t* children, double* next_base) {
    if (pf_sync(h)) return -1;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    const DState st = state_slot(h->A, c.cur);
    const long long Np = h->Np;
    const int n = h->n;
    if (w_post) HIPCHK(hipMemcpy(w_post, st.w_post, Np * 8, hipMemcpyDeviceToHost));
    if (w_pilot) HIPCHK(hipMemcpy(w_pilot, st.w_pilot, Np * 8, hipMemcpyDeviceToHost));
    if (next_base) HIPCHK(hipMemcpy(next_base, st.next_base, Np * 8, hipMemcpyDeviceToHost));
    if (heights) {
        std::vector<double> tmp((size_t)(n - 1) * Np);
        HIPCHK(hipMemcpy(tmp.data(), st.S, tmp.size() * 8, hipMemcpyDeviceToHost));
        for (long long p = 0; p < Np; ++p)
            for (int r = 0; r < n - 1; ++r) heights[p * (n - 1) + r] = tmp[(size_t)r * Np + p];
    }
    if (children) {
        std::vector<int8_t> tmp((size_t)2 * (n - 1) * Np);
        HIPCHK(hipMemcpy(tmp.data(), st.C, tmp.size(), hipMemcpyDeviceToHost));
        for (long long p = 0; p < Np; ++p)
            for (int k = 0; k < 2 * (n - 1); ++k) children[p * 2 * (n - 1) + k] = tmp[(size_t)k * Np + p];
    }
    return 0;
}

// Philox4x32-10 on the host (the same stream definition as philox_uniform in pf_device.h): the final one-particle draw
// needs a single uniform
static double philox_uniform_host(unsigned long long seed, unsigned slot, unsigned stream, unsigned long long draw) {
    unsigned c0 = (unsigned)draw, c1 = (unsigned)(draw >> 32), c2 = slot, c3 = stream;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned long long bits = (((unsigned long long)c0 << 32) | c1) >> 11;
    return ((double)bits + 0.5) * 1.1102230246251565e-16;
}

// ------------------------------------------------------------------ -arg: the history of one particle
// ParticleContainer::printTrees (pc.cpp:515-555) prints the tree-modifying events of the particle left by the final
// one-particle draw (smcsmc.cpp:395).  Here: walk back through the generations (the parent tables of all resamplings
// are kept with -arg), collect the record ranges of every ancestor, copy them out.
__global__ void k_trace_ancestry(KArgs A, int G, int slot, int* out_slot, unsigned* out_k0, unsigned* out_k1) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    long long a = slot;
    for (int g = G; g >= 0; --g) {
        out_slot[g] = (int)a;
        out_k0[g] = A.gstart[(size_t)(g % A.Gcap) * A.Np + a];
        out_k1[g] = g == G ? A.widx[a] : A.gstart[(size_t)((g + 1) % A.Gcap) * A.Np + a];
        if (g > 0) a = A.ev_parents[(size_t)(g - 1) * A.Np + a];
    }
}
__global__ void k_gather_records(KArgs A, int ngen, const int* slot, const unsigned* k0, const unsigned* k1, const long long* off,
                                 double* out) {
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= ngen) return;
    double* o = out + (size_t)off[g] * A.RS;
    for (unsigned k = k0[g]; k < k1[g]; ++k) {
        const double* rec = rec_ptr(A, slot[g], k);
        for (int j = 0; j < A.RS; ++j) *o++ = rec[j];
    }
}

// pieces of the records gathered by k_gather_records (structured models): entry i copies np[i] pieces of slot[i] from pstart[i]
__global__ void k_gather_pieces(KArgs A, int nrec, const int* slot, const unsigned* pstart, const unsigned* np, const long long* off,
                                double* out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= nrec) return;
    double* o = out + (size_t)off[i] * 3;
    for (unsigned j = 0; j < np[i]; ++j) {
        const double* q = A.plog + ((size_t)slot[i] * A.pcap + ((pstart[i] + j) % A.pcap)) * 3;
        *o++ = q[0]; *o++ = q[1]; *o++ = q[2];
    }
}

int64_t pf_sample_tree_events(pf_handle* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int64_t max_events,
                              int64_t* particle_out) {
    return pf_sample_tree_events_pops(h, kind, pos, height, desc, nullptr, nullptr, max_events, particle_out);
}

int64_t pf_sample_tree_events_pops(pf_handle* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int32_t* from_pop,
                                   int32_t* to_pop, int64_t max_events, int64_t* particle_out) {
    if (!h->A.rec_trees) { g_err = "pf_sample_tree_events: the handle was not created with tree recording (pf_params.flags bit 1)"; return -1; }
    if (pf_sync(h)) return -1;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    const long long Np = h->Np;
    // the one-particle systematic draw of resample(..., NULL, 1): the particle whose cumulative pilot weight
    // (pilotWeight(), pc.cpp:256-262) passes U * total
    std::vector<double> w(Np);
    HIPCHK(hipMemcpy(w.data(), state_slot(h->A, c.cur).w_pilot, Np * 8, hipMemcpyDeviceToHost));
    double total = 0.0;
    for (double v : w) total += v;
    const double u = philox_uniform_host((unsigned long long)h->A.seed, 0xFFFFFFFFu, 1, (unsigned long long)c.n_resample);
    long long j = 0;
    double acc = 0.0;
    for (; j < Np - 1; ++j) { acc += w[j]; if (acc > u * total) break; }
    if (particle_out) *particle_out = j;
    const int G = c.gen;
    int* dslot; unsigned *dk0, *dk1; long long* doff;
    HIPCHK(hipMalloc(&dslot, (size_t)(G + 1) * 4)); HIPCHK(hipMalloc(&dk0, (size_t)(G + 1) * 4)); HIPCHK(hipMalloc(&dk1, (size_t)(G + 1) * 4));
    HIPCHK(hipMalloc(&doff, (size_t)(G + 1) * 8));
    hipLaunchKernelGGL(k_trace_ancestry, dim3(1), dim3(1), 0, h->stream, h->A, G, (int)j, dslot, dk0, dk1);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<unsigned> k0(G + 1), k1(G + 1);
    HIPCHK(hipMemcpy(k0.data(), dk0, (size_t)(G + 1) * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(k1.data(), dk1, (size_t)(G + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<long long> off(G + 1);
    long long nrec = 0;
    for (int g = 0; g <= G; ++g) { off[g] = nrec; nrec += (long long)(k1[g] - k0[g]); }
    HIPCHK(hipMemcpy(doff, off.data(), (size_t)(G + 1) * 8, hipMemcpyHostToDevice));
    const int RS = h->A.RS;
    double* drec;
    HIPCHK(hipMalloc(&drec, std::max<size_t>(1, (size_t)nrec * RS) * 8));
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((G + 1 + 255) / 256)), dim3(256), 0, h->stream, h->A, G + 1, dslot, dk0, dk1, doff, drec);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> rec((size_t)nrec * RS);
    if (nrec) HIPCHK(hipMemcpy(rec.data(), drec, rec.size() * 8, hipMemcpyDeviceToHost));
    // structured models: the migrations and the coalescence of an update are in its pieces (pf_mp.h PLog)
    std::vector<double> pieces;
    std::vector<long long> poff((size_t)nrec + 1, 0);
    if (h->P > 1 && nrec) {
        std::vector<int> gslot(G + 1);
        HIPCHK(hipMemcpy(gslot.data(), dslot, (size_t)(G + 1) * 4, hipMemcpyDeviceToHost));
        std::vector<int> rslot((size_t)nrec);
        std::vector<unsigned> rstart((size_t)nrec), rnp((size_t)nrec);
        for (int g = 0; g <= G; ++g)
            for (long long r = off[g]; r < off[g] + (long long)(k1[g] - k0[g]); ++r) rslot[(size_t)r] = gslot[g];
        for (long long r = 0; r < nrec; ++r) {
            const double* q = &rec[(size_t)r * RS];
            unsigned long long meta, ref;
            memcpy(&meta, &q[4], 8);
            memcpy(&ref, &q[3], 8);
            const int type = (int)(meta & 0xff);
            const bool has = type == 0 || type == 2;
            rstart[(size_t)r] = has ? (unsigned)(ref & 0xffffffffu) : 0u;
            rnp[(size_t)r] = has ? (unsigned)(ref >> 32) : 0u;
            poff[(size_t)r + 1] = poff[(size_t)r] + rnp[(size_t)r];
        }
        const long long npieces = poff[(size_t)nrec];
        int* d_s; unsigned *d_p0, *d_np; long long* d_off; double* d_out;
        HIPCHK(hipMalloc(&d_s, (size_t)nrec * 4)); HIPCHK(hipMalloc(&d_p0, (size_t)nrec * 4)); HIPCHK(hipMalloc(&d_np, (size_t)nrec * 4));
        HIPCHK(hipMalloc(&d_off, (size_t)nrec * 8)); HIPCHK(hipMalloc(&d_out, std::max<size_t>(1, (size_t)npieces * 3) * 8));
        HIPCHK(hipMemcpy(d_s, rslot.data(), (size_t)nrec * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_p0, rstart.data(), (size_t)nrec * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_np, rnp.data(), (size_t)nrec * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_off, poff.data(), (size_t)nrec * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_gather_pieces, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, h->stream, h->A, (int)nrec, d_s, d_p0, d_np, d_off, d_out);
        HIPCHK(hipStreamSynchronize(h->stream));
        pieces.resize((size_t)npieces * 3);
        if (npieces) HIPCHK(hipMemcpy(pieces.data(), d_out, pieces.size() * 8, hipMemcpyDeviceToHost));
        hipFree(d_s); hipFree(d_p0); hipFree(d_np); hipFree(d_off); hipFree(d_out);
    }
    hipFree(dslot); hipFree(dk0); hipFree(dk1); hipFree(doff); hipFree(drec);
    // last position first; within an update the R line precedes the C line, which precedes the M lines of the walk that
    // led to it, latest first (the reference prepends every event to one list and prints from its head, pc.cpp:527-551)
    int64_t nout = 0;
    auto emit = [&](int kd, double x, double t, unsigned d, int from, int to) {
        if (nout < max_events) {
            if (kind) kind[nout] = kd;
            if (pos) pos[nout] = x;
            if (height) height[nout] = t;
            if (desc) desc[nout] = d;
            if (from_pop) from_pop[nout] = from;
            if (to_pop) to_pop[nout] = to;
        }
        ++nout;
    };
    const unsigned full = (1u << h->n) - 1u;
    for (long long r = nrec - 1; r >= 0; --r) {
        const double* q = &rec[(size_t)r * RS];
        unsigned long long meta;
        memcpy(&meta, &q[4], 8);
        const int type = (int)(meta & 0xff);
        if (type != 0 && type != 2) continue;
        const unsigned cut = (unsigned)((meta >> 32) & 0xffff), below = (unsigned)((meta >> 48) & 0xffff);
        if (type == 0) emit(0, q[1], q[2], cut, -1, -1);
        if (h->P == 1) { emit(1, q[1], q[3], below, 0, -1); continue; }
        // the floating lineage is the cut branch (an update) or leaf i on its way into the partial tree (initial tree:
        // n_eff = i); the other active lineage is the root's own, above everything that is not the floating lineage
        const int leaf = (int)((meta >> 24) & 0xff);
        const unsigned fl = type == 0 ? cut : (1u << leaf);
        const unsigned rt = type == 0 ? (full & ~cut) : ((1u << leaf) - 1u);
        for (long long j = poff[(size_t)r + 1] - 1; j >= poff[(size_t)r]; --j) {
            long long tag;
            memcpy(&tag, &pieces[(size_t)j * 3], 8);
            const int pop = (int)(tag & 0xff), kd = (int)((tag >> 8) & 0xff), to = (int)((tag >> 16) & 0xff);
            const double t1 = pieces[(size_t)j * 3 + 2];
            if (kd & 1) emit(1, q[1], t1, below, pop, -1);
            if (kd & 2) emit(2, q[1], t1, (kd & 4) ? rt : fl, pop, to);
        }
    }
    return nout;
}

int pf_set_timing(pf_handle* h, int period) {
    h->timing_period = period;
    if (period > 0 && h->span_overhead_ms == 0) {
        // A span is two event records around one launch; what the pair measures with nothing in between is not kernel
        // time.  Median of 64 empty spans, subtracted from every span, so that the per-launch figure agrees with the
        // kernel durations a profiler reports.
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        std::vector<float> empty;
        for (int i = 0; i < 64; ++i) {
            hipEvent_t a = get_event(h), b = get_event(h);
            hipEventRecord(a, h->stream); hipEventRecord(b, h->stream);
            HIPCHK(hipEventSynchronize(b));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, a, b));
            empty.push_back(ms);
            h->ev_pool.push_back(a); h->ev_pool.push_back(b);
        }
        std::sort(empty.begin(), empty.end());
        h->span_overhead_ms = empty[empty.size() / 2];
    }
    return 0;
}

// profiling builds (-DPF_STAMPS): wall-clock stamps (100 MHz) of the phases of the extend workgroups, one set per row and
// wavefront; a regular build records nothing
int pf_test_search_lut(const double* tab, int32_t n, uint8_t* lut, int32_t* kbase) {
    int kb = 0;
    if (n < 1 || n > PF_EMAX || !lut_build(tab, n, lut, &kb)) return 0;
    *kbase = kb;
    return 1;
}

int pf_debug_stamps(pf_handle* h, int64_t rows, uint64_t* out) {
    HIPCHK(hipSetDevice(h->device));
    if (rows > 0 && !out) {
        unsigned long long* buf = nullptr;
        if (dalloc(h, &buf, (size_t)rows * h->A.nc * PF_STAMP_W)) return -1;
        HIPCHK(hipMemset(buf, 0, (size_t)rows * h->A.nc * PF_STAMP_W * 8));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->A.stamps = buf; h->A.stamp_rows = rows;
        return 0;
    }
    if (!h->A.stamps) { g_err = "pf_debug_stamps: not enabled"; return -1; }
    if (pf_sync(h)) return -1;
    HIPCHK(hipMemcpy(out, h->A.stamps, (size_t)std::min<long long>(rows, h->A.stamp_rows) * h->A.nc * PF_STAMP_W * 8, hipMemcpyDeviceToHost));
    return 0;
}

int pf_get_kernel_time(pf_handle* h, int k, double* ms, int64_t* launches) {
    if (k < 0 || k > 3) return -1;
    if (pf_sync(h)) return -1;
    // mean duration of the timed launches, scaled to all launches of this class
    double mean = h->k_timed[k] ? h->k_ms[k] / (double)h->k_timed[k] : 0.0;
    if (ms) *ms = mean * (double)h->k_launches[k];
    if (launches) *launches = h->k_launches[k];
    return 0;
}

// Measurement aid: steps [first_step, first_step + n_steps) of the next pf_run / pf_run_many calls led by this handle run the traced
// instance of the row kernel (k_sweep4t: at most four haplotypes, no focused sampling; other shapes ignore the request).
int pf_set_wg_trace(pf_handle* h, int64_t first_step, int32_t n_steps) {
    HIPCHK(hipSetDevice(h->device));
    if (pf_sync(h)) return -1;
    if (h->d_trace) { hipFree(h->d_trace); h->d_trace = nullptr; }
    h->trace_t0 = (int)std::max<int64_t>(0, first_step);
    h->trace_n = std::max(0, n_steps);
    h->trace_stride = 0; h->trace_words = 0;
    return 0;
}

// The trace: four 64-bit words per workgroup slot of every traced step -- start and end (100 MHz clock; 0 0: slot not used by that step's
// grid), HW_ID | XCC_ID << 32, index within the chunk | chunk << 32.  Returns the number of words (0: nothing traced yet); copies at most
// cap_words of them; info = {steps, workgroup slots per step}.
int64_t pf_get_wg_trace(pf_handle* h, uint64_t* out, int64_t cap_words, int32_t* info) {
    if (hipSetDevice(h->device) != hipSuccess || pf_sync(h)) return -1;
    if (info) { info[0] = h->d_trace ? h->trace_n : 0; info[1] = h->trace_stride; }
    if (!h->d_trace) return 0;
    const size_t nw = std::min<size_t>(h->trace_words, (size_t)std::max<int64_t>(0, cap_words));
    if (out && nw && hipMemcpy(out, h->d_trace, nw * 8, hipMemcpyDeviceToHost) != hipSuccess) { g_err = "copy of the workgroup trace failed"; return -1; }
    return (int64_t)h->trace_words;
}

int pf_get_delay_stats(pf_handle* h, int64_t* n_forced, int32_t* peak_pending) {
    if (pf_sync(h)) return -1;
    Ctrl c;
    HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    if (n_forced) *n_forced = (int64_t)c.n_delay_evict;
    if (peak_pending) *peak_pending = c.delay_peak;
    return 0;
}

int pf_get_stats(pf_handle* h, int64_t* n_records, int64_t* state_bytes, int64_t* n_resamples) {
    if (pf_sync(h)) return -1;
    if (n_records) {
        std::vector<unsigned> w(h->Np);
        HIPCHK(hipMemcpy(w.data(), h->A.widx, h->Np * 4, hipMemcpyDeviceToHost));
        long long t = 0;
        for (unsigned v : w) t += v;
        *n_records = t;
    }
    if (state_bytes) *state_bytes = (int64_t)(h->n - 1) * 8 + 2 * (h->n - 1) + 5 * 8 + 4 + 8 + 8 + 4;
    if (n_resamples) {
        Ctrl c;
        HIPCHK(hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
        *n_resamples = c.n_resample;
    }
    return 0;
}

// ------------------------------------------------------------------ unit-level test entry points
static int test_setup(int device) {
    if (pf_device_count() <= 0) { g_err = "no HIP device"; return -1; }
    HIPCHK(hipSetDevice(device));
    return 0;
}

int pf_test_math(const double* x, int64_t n, double* oe, double* ol, double* of, int device) {
    if (test_setup(device)) return -1;
    double *dx, *de, *dl, *df;
    HIPCHK(hipMalloc(&dx, n * 8)); HIPCHK(hipMalloc(&de, n * 8)); HIPCHK(hipMalloc(&dl, n * 8)); HIPCHK(hipMalloc(&df, n * 8));
    HIPCHK(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, (long long)n, de, dl, df);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(oe, de, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ol, dl, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(of, df, n * 8, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(de); hipFree(dl); hipFree(df);
    return 0;
}

int pf_test_div(const double* a, const double* b, int64_t n, double* out, int device) {
    if (test_setup(device)) return -1;
    double *da, *db, *dout;
    HIPCHK(hipMalloc(&da, n * 8)); HIPCHK(hipMalloc(&db, n * 8)); HIPCHK(hipMalloc(&dout, n * 8));
    HIPCHK(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, (long long)n, dout);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dout);
    return 0;
}

int pf_test_uniform(uint64_t seed, uint32_t slot, uint32_t stream, uint64_t first_draw, int64_t n, double* out, int device) {
    if (test_setup(device)) return -1;
    double* d;
    HIPCHK(hipMalloc(&d, n * 8));
    hipLaunchKernelGGL(k_test_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned long long)seed, slot, stream,
                       (unsigned long long)first_draw, (long long)n, d);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d, n * 8, hipMemcpyDeviceToHost));
    hipFree(d);
    return 0;
}

// canonical sum / scan / systematic table through the production kernels (k_partials + k_decide)
static int test_reduce_impl(const double* x, int64_t n, double* out_sum, double* out_scan, double u, int32_t* lo, int device) {
    if (test_setup(device)) return -1;
    pf_model m;
    memset(&m, 0, sizeof(m));
    double ct = 0.0, ps = 1e4, lag = 1e30;
    int rf = 3;
    m.n_epochs = 1; m.n_pops = 1; m.nsam = 2; m.loci_length = 1e6; m.mutation_rate = 1e-8; m.recombination_rate = 1e-8;
    m.change_times = &ct; m.pop_sizes = &ps; m.record_flags = &rf; m.lags = &lag;
    pf_params p;
    memset(&p, 0, sizeof(p));
    p.np = n; p.ess_fraction = lo ? 2.0 : 0.0; p.seed = 1; p.max_trace_events = 0;   // ESS < 2N always -> forces the table
    pf_handle* h = create_impl(&m, &p, device, 8, 4);
    if (!h) return -1;
    int rc = 0;
    double s0 = 0.0, l0 = 1.0; int8_t st0 = 1; int8_t al[2] = {-1, -1}; int lim0 = 0;
    pf_segments sg = {1, &s0, &l0, &st0, al, &lim0};
    rc |= pf_load_segments(h, &sg);
    rc |= pf_init_prior(h, 0.0);
    if (!rc) {
        hipMemcpyAsync(h->A.st0.w_post, x, n * 8, hipMemcpyHostToDevice, h->stream);
        hipMemcpyAsync(h->A.st0.w_pilot, x, n * 8, hipMemcpyHostToDevice, h->stream);
        hipLaunchKernelGGL(k_partials, dim3(h->nblocks), dim3(PF_BS), 0, h->stream, h->A);
        // override u by running k_decide in mode 0 and then patching: simpler -- write u after the fact
        hipLaunchKernelGGL(k_decide, dim3(h->nblocks + 1), dim3(PF_BS), 0, h->stream, h->A, (long long)0, lo ? 0 : 1, no_windows(h), h->nblocks);
        hipStreamSynchronize(h->stream);
        Ctrl c;
        hipMemcpy(&c, h->A.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost);
        if (out_sum) *out_sum = c.T;
        if (out_scan) {
            std::vector<double> s1(n), off((n + 63) / 64);
            hipMemcpy(s1.data(), h->A.scan1, n * 8, hipMemcpyDeviceToHost);
            hipMemcpy(off.data(), h->A.chunk_off, off.size() * 8, hipMemcpyDeviceToHost);
            for (int64_t i = 0; i < n; ++i) out_scan[i] = off[i >> 6] + s1[i];
        }
        if (lo) {
            (void)u;
            hipMemcpy(lo, h->A.lo, (n + 1) * 4, hipMemcpyDeviceToHost);
        }
    }
    pf_destroy(h);
    return rc ? -1 : 0;
}

int pf_test_reduce(const double* x, int64_t n, double* out_sum, double* out_incl_scan, int device) {
    return test_reduce_impl(x, n, out_sum, out_incl_scan, 0.0, nullptr, device);
}

int pf_test_systematic(const double* pilot, int64_t n, double u, int32_t* lo, int device) {
    // the production kernel draws u from the resampler stream (seed 1, event 0); the caller
    // obtains the same u through pf_test_uniform(1, 0xFFFFFFFF, 1, 0, ...)
    return test_reduce_impl(pilot, n, nullptr, nullptr, u, lo, device);
}

int pf_load_lookahead(pf_handle* h, const pf_lookahead* la) {
    HIPCHK(hipSetDevice(h->device));
    if (la->level < 0 || la->level > 4) { g_err = "-apf must be in 0..4"; return -1; }
    if (la->n != h->n_segs) { g_err = "pf_load_lookahead: one look-ahead record per segment expected"; return -1; }
    if (la->level == 0) { h->A.apf = 0; return 0; }
    const long long S = la->n;
    const int n = h->n, D = la->max_doubletons, Q = la->n_quantiles;
    KArgs& A = h->A;
    double *fsd, *rmr, *ddist, *split, *q, *tbl; int8_t *unph, *didx, *sal; int *nd, *sk;
    int rc = 0;
    rc |= dalloc(h, &fsd, (size_t)S * n); rc |= dalloc(h, &rmr, (size_t)S * n); rc |= dalloc(h, &unph, (size_t)S * n);
    rc |= dalloc(h, &nd, S); rc |= dalloc(h, &didx, (size_t)S * D * 4); rc |= dalloc(h, &ddist, (size_t)S * D * 2);
    rc |= dalloc(h, &split, S); rc |= dalloc(h, &sal, (size_t)S * n); rc |= dalloc(h, &sk, S);
    rc |= dalloc(h, &q, Q); rc |= dalloc(h, &tbl, (size_t)n * Q);
    if (!A.st0.lookahead) rc |= dalloc(h, &A.st0.lookahead, (size_t)A.nslots * h->Np);
    if (rc) return -1;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(fsd, la->first_singleton_distance, (size_t)S * n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(rmr, la->relative_mutation_rate, (size_t)S * n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(unph, la->is_singleton_unphased, (size_t)S * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(nd, la->n_doubletons, (size_t)S * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(didx, la->doubleton_idx, (size_t)S * D * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ddist, la->doubleton_dist, (size_t)S * D * 2 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(split, la->first_split_distance, (size_t)S * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sal, la->split_alleles, (size_t)S * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sk, la->split_count, (size_t)S * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(q, la->quantiles, (size_t)Q * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(tbl, la->tbl_lengths, (size_t)n * Q * 8, hipMemcpyHostToDevice));
    std::vector<double> ones((size_t)A.nslots * h->Np, 1.0);
    HIPCHK(hipMemcpy(A.st0.lookahead, ones.data(), ones.size() * 8, hipMemcpyHostToDevice));
    A.apf = la->level; A.la_D = D; A.la_Q = Q;
    A.la_fsd = fsd; A.la_rmr = rmr; A.la_unph = unph; A.la_nd = nd; A.la_didx = didx; A.la_ddist = ddist;
    A.la_split = split; A.la_salleles = sal; A.la_sk = sk; A.la_q = q; A.la_tbl = tbl;
    A.la_mean_tbl = la->mean_total_branch_length;
    h->smem_la = smem_bytes_la(n, h->E);
    if (h->smem_la > 64 * 1024) hipFuncSetAttribute((const void*)k_lookahead, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_la);
    return 0;
}

int pf_terminal_branch_quantiles(const pf_model* m, uint64_t seed, int64_t n_trees, const double* quantiles, int32_t nq,
                                 double* lengths_out, double* mean_total_out, int device) {
    if (test_setup(device)) return -1;
    if (m->n_pops < 1 || m->n_pops > PF_PMAX || m->nsam < 2 || m->nsam > PF_NMAX || m->n_epochs < 1 || m->n_epochs > PF_EMAX || n_trees < 1) {
        g_err = "pf_terminal_branch_quantiles: unsupported model";
        return -1;
    }
    const int E = m->n_epochs, n = m->nsam, P = m->n_pops;
    KArgs A;
    memset(&A, 0, sizeof(A));
    A.E = E; A.n = n; A.P = P; A.L = m->loci_length; A.mu = m->mutation_rate; A.rho = m->recombination_rate; A.mcap = PF_MMAX;
    double *dT, *dI, *dh, *dl; int *dRF, *derr;
    std::vector<void*> mp_allocs;
    HIPCHK(hipMalloc(&dT, E * 8)); HIPCHK(hipMalloc(&dI, E * 8)); HIPCHK(hipMalloc(&dRF, E * 4)); HIPCHK(hipMalloc(&derr, 4));
    HIPCHK(hipMemset(derr, 0, 4));
    HIPCHK(hipMalloc(&dh, (size_t)n * n_trees * 8)); HIPCHK(hipMalloc(&dl, (size_t)n_trees * 8));
    std::vector<double> inv2N(E);
    for (int e = 0; e < E; ++e) inv2N[e] = 1.0 / (2.0 * m->pop_sizes[(size_t)e * P]);
    std::vector<int> rf(E, 3);
    HIPCHK(hipMemcpy(dT, m->change_times, E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dI, inv2N.data(), E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dRF, rf.data(), E * 4, hipMemcpyHostToDevice));
    double* dHc;
    HIPCHK(hipMalloc(&dHc, E * 8));
    {
        const std::vector<double> Hc = cumulative_intensity(m->change_times, inv2N, E);
        HIPCHK(hipMemcpy(dHc, Hc.data(), E * 8, hipMemcpyHostToDevice));
    }
    A.T = dT; A.inv2N = dI; A.Hc = dHc; A.recflags = dRF;
    int tbl_err = 0;
    if (P > 1) {
        MpTables tb;
        if (build_mp_tables(m, tb) || upload_mp_tables(tb, A, mp_allocs)) return -1;
        const size_t smem = pf_mp_smem_bytes(n, E, P, PF_MMAX);
        if (pf_mp_prepare(smem, PF_MMAX)) { g_err = "pf_terminal_branch_quantiles: the local-tree state does not fit the LDS"; return -1; }
        pf_mp_launch_tbl(A, (unsigned long long)seed, (long long)n_trees, dh, dl, derr, smem, 0);
    } else {
        const size_t smem = smem_bytes(n, E);
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)k_tbl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k_tbl, dim3((unsigned)((n_trees + PF_BS - 1) / PF_BS)), dim3(PF_BS), smem, 0, A, (unsigned long long)seed,
                           (long long)0, (long long)n_trees, dh, dl);
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(&tbl_err, derr, 4, hipMemcpyDeviceToHost));
    std::vector<double> hh((size_t)n * n_trees), ll(n_trees);
    HIPCHK(hipMemcpy(hh.data(), dh, hh.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ll.data(), dl, ll.size() * 8, hipMemcpyDeviceToHost));
    hipFree(dT); hipFree(dI); hipFree(dHc); hipFree(dRF); hipFree(dh); hipFree(dl); hipFree(derr);
    for (void* q : mp_allocs) hipFree(q);
    if (tbl_err) {
        g_err = tbl_err == 1 ? "too many migration events on one local tree" : "No final coalescence event was sampled!";
        return -1;
    }
    for (int i = 0; i < n; ++i) {
        double* row = hh.data() + (size_t)i * n_trees;
        std::sort(row, row + n_trees);
        for (int q = 0; q < nq; ++q) lengths_out[i * nq + q] = row[(long long)(quantiles[q] * (double)n_trees)];
    }
    double total = 0.0;
    for (long long r = 0; r < n_trees; ++r) total += ll[r];      // serial, in tree order (smcsmc.cpp:145)
    *mean_total_out = total / (double)n_trees;
    return 0;
}

// ------------------------------------------------------------------ lag calibration (host driver)
// calculate_median_survival_distances (smcsmc.cpp:169-263).  The reference grows its sample one
// tree at a time with a shared RNG; here trees are simulated in fixed batches of PF_CAL_BATCH
// independent Philox streams until every epoch has min_events samples (or max_trees is reached),
// then the same median / fallback rules are applied (smcsmc.cpp:235-262).
#define PF_CAL_BATCH 16384

int pf_median_survival(const pf_model* m, uint64_t seed, int32_t min_events, int64_t max_trees, double* median_out,
                       int64_t* trees_used, int device) {
    if (test_setup(device)) return -1;
    if (m->n_pops < 1 || m->n_pops > PF_PMAX || m->nsam < 2 || m->nsam > PF_NMAX || m->n_epochs < 1 || m->n_epochs > PF_EMAX) {
        g_err = "pf_median_survival: unsupported model";
        return -1;
    }
    const int E = m->n_epochs, n = m->nsam, P = m->n_pops;
    KArgs A;
    memset(&A, 0, sizeof(A));
    A.E = E; A.n = n; A.P = P; A.L = m->loci_length; A.mu = m->mutation_rate; A.rho = m->recombination_rate; A.mcap = PF_MMAX;
    double *dT, *dI; int *dRF, *dep, *derr; double* ddist;
    std::vector<void*> mp_allocs;
    HIPCHK(hipMalloc(&dT, E * 8)); HIPCHK(hipMalloc(&dI, E * 8)); HIPCHK(hipMalloc(&dRF, E * 4)); HIPCHK(hipMalloc(&derr, 4));
    HIPCHK(hipMemset(derr, 0, 4));
    // The stopping rule is evaluated batch by batch (D8), but PF_CAL_GROUP batches are simulated per launch: a batch of
    // 16 384 trees fills a quarter of the SIMDs and takes as long as four.  Batches behind the one at which the rule
    // fires are discarded, so the result is that of one launch per batch.
    constexpr int PF_CAL_GROUP = 4;
    const size_t per_batch = (size_t)PF_CAL_BATCH * (n - 1);
    HIPCHK(hipMalloc(&dep, PF_CAL_GROUP * per_batch * 4)); HIPCHK(hipMalloc(&ddist, PF_CAL_GROUP * per_batch * 8));
    std::vector<double> inv2N(E);
    for (int e = 0; e < E; ++e) inv2N[e] = 1.0 / (2.0 * m->pop_sizes[(size_t)e * P]);
    std::vector<int> rf(E, 3);
    HIPCHK(hipMemcpy(dT, m->change_times, E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dI, inv2N.data(), E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dRF, rf.data(), E * 4, hipMemcpyHostToDevice));
    double* dHc;
    HIPCHK(hipMalloc(&dHc, E * 8));
    {
        const std::vector<double> Hc = cumulative_intensity(m->change_times, inv2N, E);
        HIPCHK(hipMemcpy(dHc, Hc.data(), E * 8, hipMemcpyHostToDevice));
    }
    A.T = dT; A.inv2N = dI; A.Hc = dHc; A.recflags = dRF;
    if (P > 1) {
        MpTables tb;
        if (build_mp_tables(m, tb) || upload_mp_tables(tb, A, mp_allocs)) return -1;
    }
    std::vector<std::vector<double>> surv(E);
    std::vector<int> hep(PF_CAL_GROUP * per_batch);
    std::vector<double> hdist(PF_CAL_GROUP * per_batch);
    long long trees = 0;
    const size_t smem = P > 1 ? pf_mp_smem_bytes(n, E, P, PF_MMAX) : smem_bytes(n, E);
    if (P > 1 && pf_mp_prepare(smem, PF_MMAX)) { g_err = "pf_median_survival: the local-tree state does not fit the LDS"; return -1; }
    const size_t smem_cal = smem + (size_t)(2 * PF_EPAD + E) * 8;          // + the padded epoch tables of the register path
    if (P == 1 && smem_cal > 64 * 1024) {
        hipFuncSetAttribute((const void*)k_calibrate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_cal);
        hipFuncSetAttribute((const void*)k_calibrate<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_cal);
        hipFuncSetAttribute((const void*)k_calibrate<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_cal);
    }
    int cal_err = 0;
    auto finished = [&]() {
        int not_done = 0;
        for (int e = 0; e < E; ++e) not_done += (int)surv[e].size() < min_events;
        return not_done == 0 || trees >= max_trees;
    };
    while (!finished()) {
        const long long nrep = (long long)PF_CAL_GROUP * PF_CAL_BATCH;
        if (P > 1)
            pf_mp_launch_calibrate(A, (unsigned long long)seed, trees, nrep, dep, ddist, derr, smem, 0);
        else
        {
            const dim3 grid((unsigned)(nrep / PF_BS)), blk(PF_BS);
            if (n <= 4) hipLaunchKernelGGL(k_calibrate<4>, grid, blk, smem_cal, 0, A, (unsigned long long)seed, trees, nrep, dep, ddist);
            else if (n <= 8) hipLaunchKernelGGL(k_calibrate<8>, grid, blk, smem_cal, 0, A, (unsigned long long)seed, trees, nrep, dep, ddist);
            else hipLaunchKernelGGL(k_calibrate<0>, grid, blk, smem_cal, 0, A, (unsigned long long)seed, trees, nrep, dep, ddist);
        }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(&cal_err, derr, 4, hipMemcpyDeviceToHost));
        if (cal_err) break;
        HIPCHK(hipMemcpy(hep.data(), dep, hep.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hdist.data(), ddist, hdist.size() * 8, hipMemcpyDeviceToHost));
        for (int b = 0; b < PF_CAL_GROUP && !finished(); ++b) {
            for (size_t i = b * per_batch; i < (b + 1) * per_batch; ++i)
                if (hdist[i] >= 0) surv[hep[i]].push_back(hdist[i]);
            trees += PF_CAL_BATCH;
        }
    }
    hipFree(dT); hipFree(dI); hipFree(dHc); hipFree(dRF); hipFree(dep); hipFree(ddist); hipFree(derr);
    for (void* q : mp_allocs) hipFree(q);
    if (cal_err) {
        g_err = cal_err == 1 ? "too many migration events on one local tree"
              : cal_err == 3 ? "No final coalescence event was sampled!" : "structured-model genealogy update failed";
        return -1;
    }
    if (trees_used) *trees_used = trees;
    // smcsmc.cpp:235-262
    double earliest = -1;
    for (int e = 0; e < E; ++e) {
        std::sort(surv[e].begin(), surv[e].end());
        int median_idx = ((int)surv[e].size() - 1) / 2;
        if (median_idx < 10) median_out[e] = -1;
        else {
            median_out[e] = surv[e][median_idx];
            if (earliest < 0) earliest = median_out[e];
        }
    }
    for (int e = 0; e < E; ++e)
        if (median_out[e] < 0) median_out[e] = e > 0 ? median_out[e - 1] : earliest;
    return 0;
}

// Synthetic data for `nchunks` independent chunks of the model's length (one population): site positions and carrier
// masks per chunk (k_simulate).  n_sites[c] < 0: more than max_sites sites (the first max_sites are valid).
int pf_simulate_sites(const pf_model* m, uint64_t seed, int32_t nchunks, int64_t max_sites, double* pos, uint32_t* masks,
                      int64_t* n_sites, int device) {
    if (test_setup(device)) return -1;
    if (m->n_pops < 1 || m->n_pops > PF_PMAX || m->nsam < 2 || m->nsam > PF_NMAX || m->n_epochs < 1 || m->n_epochs > PF_EMAX || nchunks < 1 || max_sites < 1) {
        g_err = "pf_simulate_sites: 1..4 populations, 2..16 haplotypes, 1..64 epochs";
        return -1;
    }
    const int E = m->n_epochs, n = m->nsam, P = m->n_pops;
    KArgs A;
    memset(&A, 0, sizeof(A));
    A.E = E; A.n = n; A.P = P; A.L = m->loci_length; A.mu = m->mutation_rate; A.rho = m->recombination_rate; A.mcap = PF_MMAX;
    // every exit path frees what was allocated so far
    struct Bufs { std::vector<void*> v; ~Bufs() { for (void* q : v) hipFree(q); } } bufs;
    auto dmalloc = [&](auto** q, size_t bytes) { void* r = nullptr; hipError_t e = hipMalloc(&r, bytes); if (e == hipSuccess) bufs.v.push_back(r); *q = (std::remove_reference_t<decltype(**q)>*)r; return e; };
    double *dT, *dI, *dHc, *dpos; int* dRF; unsigned* dmask; long long* dn;
    HIPCHK(dmalloc(&dT, E * 8)); HIPCHK(dmalloc(&dI, E * 8)); HIPCHK(dmalloc(&dHc, E * 8)); HIPCHK(dmalloc(&dRF, E * 4));
    HIPCHK(dmalloc(&dpos, (size_t)nchunks * max_sites * 8)); HIPCHK(dmalloc(&dmask, (size_t)nchunks * max_sites * 4));
    HIPCHK(dmalloc(&dn, (size_t)nchunks * 8));
    std::vector<double> inv2N(E);
    for (int e = 0; e < E; ++e) inv2N[e] = 1.0 / (2.0 * m->pop_sizes[(size_t)e * P]);
    std::vector<int> rf(E, 3);
    const std::vector<double> Hc = cumulative_intensity(m->change_times, inv2N, E);
    HIPCHK(hipMemcpy(dT, m->change_times, E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dI, inv2N.data(), E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dHc, Hc.data(), E * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dRF, rf.data(), E * 4, hipMemcpyHostToDevice));
    A.T = dT; A.inv2N = dI; A.Hc = dHc; A.recflags = dRF;
    std::vector<void*> mp_allocs;
    int* derr = nullptr;
    int rc = 0;
    if (P > 1) {
        // structured model: the LDS-tree walk of pf_mp.h, one lane per chunk
        MpTables tb;
        const int up = build_mp_tables(m, tb) || upload_mp_tables(tb, A, mp_allocs);
        for (void* q : mp_allocs) bufs.v.push_back(q);
        if (up) return -1;
        HIPCHK(dmalloc(&derr, 4));
        HIPCHK(hipMemset(derr, 0, 4));
        const size_t smem = pf_mp_smem_bytes(n, E, P, PF_MMAX);
        if (pf_mp_prepare(smem, PF_MMAX)) { g_err = "pf_simulate_sites: the local-tree state does not fit the LDS"; return -1; }
        pf_mp_launch_simulate(A, (unsigned long long)seed, (int)nchunks, (long long)max_sites, dpos, dmask, dn, derr, smem, 0);
    } else {
        const size_t smem = smem_bytes(n, E);
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)k_simulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k_simulate, dim3((unsigned)((nchunks + PF_BS - 1) / PF_BS)), dim3(PF_BS), smem, 0, A, (unsigned long long)seed,
                           (int)nchunks, (long long)max_sites, dpos, dmask, dn);
    }
    rc = check_launch("k_simulate");
    if (!rc && hipDeviceSynchronize() != hipSuccess) { g_err = "k_simulate failed"; rc = -1; }
    if (!rc && derr) {
        int herr = 0;
        HIPCHK(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
        if (herr) { g_err = herr == 1 ? "too many migration events on one local tree" : (herr == 3 ? "No final coalescence event was sampled!" : "structured simulation: internal error"); rc = -1; }
    }
    if (!rc) {
        std::vector<long long> hn(nchunks);
        HIPCHK(hipMemcpy(hn.data(), dn, (size_t)nchunks * 8, hipMemcpyDeviceToHost));
        for (int c = 0; c < nchunks; ++c) {
            n_sites[c] = hn[c];
            const long long k = std::min<long long>(std::llabs(hn[c]), max_sites);
            HIPCHK(hipMemcpy(pos + (size_t)c * max_sites, dpos + (size_t)c * max_sites, (size_t)k * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(masks + (size_t)c * max_sites, dmask + (size_t)c * max_sites, (size_t)k * 4, hipMemcpyDeviceToHost));
        }
    }
    return rc;
}
