// smcsmc_amd/csrc/pf_pipe.h -- what the row pipeline's kernels share across translation units (pf_hip.hip: k_pipe, k_sweep and
// the bookkeeping / ledger / count roles; pf_mp.hip: the extend role of the structured models): the decision on a finished
// row taken redundantly by every workgroup, the offspring offsets in closed form, the plan of a step, the per-chunk table
// of k_sweep.  Workgroup size is a template parameter here (PF_BS differs between the two units).
#pragma once
#include "pf_device.h"
#include "pf_types.h"

#define PF_PIPE_BS 256          // largest workgroup that runs the pipeline's prologue (sizes the LDS carve-out)

// A fresh, opaque handle on the same argument block in the constant address space: loads through it cannot be merged with
// earlier ones nor hoisted above this point, so what a phase of a long kernel needs from the block is loaded in that phase
// instead of being held in (and spilled from) scalar registers across the phases before it.  The by-value form of the
// block (kernels that take KArgs as an argument) is returned as it is.
__device__ __forceinline__ KArgsC& pf_reopen(KArgsC& A) { KArgsC* q = &A; asm volatile("" : "+s"(q)); return *q; }
__device__ __forceinline__ const KArgs& pf_reopen(const KArgs& A) { return A; }

// ---- decision on a finished row, made redundantly by every workgroup that needs it (single-launch pipeline) --------
// normalize_probability (pc.cpp:420-438) and the ESS test of resample (pc.cpp:247-283) from the per-wavefront partials
// the row's extend workgroups left in ring slot `slot`: the level-2 / level-3 part of the canonical radix-64 reduction,
// operation for operation what k_decide does, so T, S1, ESS, the flag and the uniform are bit-identical everywhere.
#define PF_PIPE_STAGE 16        // wavefronts of pilot scans staged per workgroup for the parent search
struct RowDecision { double T, S1, S2, ess, inv, u; int flag; };
struct PipeLds {                // carved from the dynamic LDS of k_pipe behind the epoch tables
    double* l2s;                // [ncpad] level-2 inclusive scan of the per-wavefront pilot totals
    double* pmx;                // [ncpad + 1] pmx[ch] = largest pilot prefix sum before wavefront ch (pmx[ch + 1]: up to its end)
    double* l2_post; double* l2_sq; double* l2_tot;   // [64] each
    double* wredd;              // [PF_PIPE_BS / 64]
    double* stage;              // [PF_PIPE_STAGE * 64]
    int* slo;                   // [PF_PIPE_BS]
    int* wint;                  // [3 * PF_PIPE_BS / 64]
};
__host__ __device__ inline size_t pipe_lds_doubles(int nc) {
    const size_t ncpad = ((size_t)nc + 63) / 64 * 64;
    return ncpad + (ncpad + 1) + 3 * 64 + PF_PIPE_BS / 64 + (size_t)PF_PIPE_STAGE * 64 + (PF_PIPE_BS + 3 * (PF_PIPE_BS / 64) + 1) / 2 + 2;
}
__device__ __forceinline__ PipeLds pipe_carve(double* base, int nc) {
    const size_t ncpad = ((size_t)nc + 63) / 64 * 64;
    PipeLds q;
    q.l2s = base; base += ncpad;
    q.pmx = base; base += ncpad + 1;
    q.l2_post = base; base += 64; q.l2_sq = base; base += 64; q.l2_tot = base; base += 64;
    q.wredd = base; base += PF_PIPE_BS / 64;
    q.stage = base; base += (size_t)PF_PIPE_STAGE * 64;
    q.slo = (int*)base; q.wint = q.slo + PF_PIPE_BS;
    return q;
}
__device__ __forceinline__ double pipe_chunk_offset(const PipeLds& q, int ch) {
    double run = 0.0;
    const int gq = ch / 64;
    for (int g = 0; g < gq; ++g) run = run + q.l2_tot[g];
    double off = (ch % 64 == 0) ? 0.0 : q.l2s[ch - 1];
    return run + off;
}
// every thread of the workgroup calls this (it contains barriers); WANT_TABLE: also the prefix maxima the offspring
// table / parent search need (only computed when the row resamples)
// the partials a thread needs first, requested before anything else so that their memory round trip overlaps the
// particle's own loads (what the previous launch wrote comes from another XCD's L2: about a microsecond)
// (the second set: a workgroup of BS threads reduces up to 2 BS wavefront partials -- the structured models' 64-particle
// workgroups at Np = 20 000 have 313 for 256 threads -- and the second pass of a thread would otherwise start with a
// round trip of its own; vm / vm2 are the two consecutive entries a thread of the prefix-maxima pass owns then)
struct RowPre { double vp = 0.0, vs = 0.0, vl = 0.0, vm = 0.0, last1 = 0.0, vp2 = 0.0, vs2 = 0.0, vl2 = 0.0, vm2 = 0.0; bool have = false; };
template <int BS = PF_PIPE_BS, class KA>
__device__ __forceinline__ RowPre row_preload(const KA& A, int slot) {
    const int nc = A.nc, ch = threadIdx.x;
    const int perc = (nc + BS - 1) / BS;
    RowPre r;
    r.have = true;
    if (ch < nc) {
        r.vp = A.rg_cpost[(size_t)slot * nc + ch];
        r.vs = A.rg_csq[(size_t)slot * nc + ch];
        r.vl = A.rg_cpil[(size_t)slot * nc + ch];
    }
    if (ch + BS < nc) {
        r.vp2 = A.rg_cpost[(size_t)slot * nc + ch + BS];
        r.vs2 = A.rg_csq[(size_t)slot * nc + ch + BS];
        r.vl2 = A.rg_cpil[(size_t)slot * nc + ch + BS];
    }
    // only a resampling row uses them (the prefix maxima), but then a round trip earlier
    if (perc == 1) { if (ch < nc) r.vm = A.rg_cmx1[(size_t)slot * nc + ch]; }
    else if (perc == 2) {
        if (2 * ch < nc) r.vm = A.rg_cmx1[(size_t)slot * nc + 2 * ch];
        if (2 * ch + 1 < nc) r.vm2 = A.rg_cmx1[(size_t)slot * nc + 2 * ch + 1];
    }
    r.last1 = A.ctrl->last1[slot];
    return r;
}
template <bool WANT_TABLE, int BS = PF_PIPE_BS, class KA>
__device__ __forceinline__ RowDecision decide_row(const KA& A, const PipeLds& q, int slot, long long n_res, RowPre pre = RowPre()) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = BS / 64;
    const int nc = A.nc;
    const int ng = (nc + 63) / 64;
    const double* cpost = A.rg_cpost + (size_t)slot * nc;
    const double* csq = A.rg_csq + (size_t)slot * nc;
    const double* cpil = A.rg_cpil + (size_t)slot * nc;
    const double* cmx1 = A.rg_cmx1 + (size_t)slot * nc;
    const double last_scan1 = pre.have ? pre.last1 : A.ctrl->last1[slot];
    for (int g = wave; g < ng; g += nwaves) {
        int ch = g * 64 + lane;
        const bool first = pre.have && g == wave, second = pre.have && g == wave + nwaves;
        double vp = first ? pre.vp : (second ? pre.vp2 : (ch < nc ? cpost[ch] : 0.0));
        double vs = first ? pre.vs : (second ? pre.vs2 : (ch < nc ? csq[ch] : 0.0));
        double vl = first ? pre.vl : (second ? pre.vl2 : (ch < nc ? cpil[ch] : 0.0));
        double rp = wave_tree_sum(vp);
        double rs = wave_tree_sum(vs);
        double sc = wave_hs_scan(vl, lane);
        if (ch < nc) q.l2s[ch] = sc;
        if (lane == 63) { q.l2_post[g] = rp; q.l2_sq[g] = rs; q.l2_tot[g] = sc; }
    }
    __syncthreads();
    RowDecision d;
    {
        double vp = lane < ng ? q.l2_post[lane] : 0.0;
        double vs = lane < ng ? q.l2_sq[lane] : 0.0;
        d.T = wave_tree_sum(vp);
        d.S2 = wave_tree_sum(vs);
    }
    d.S1 = pipe_chunk_offset(q, nc - 1) + last_scan1;   // inclusive scan at the last particle (= oracle incl[N-1])
    d.ess = (d.S1 * d.S1) / d.S2;
    d.flag = (d.ess < A.ess_threshold - 1e-6) ? 1 : 0;
    d.inv = 1.0 / d.T;
    d.u = d.flag ? philox_uniform(A.seed, 0xFFFFFFFFu, 1, (unsigned long long)n_res) : 0.0;
    if (WANT_TABLE && d.flag) {
        // pmx[ch] = max over wavefronts c' < ch of (chunk_off[c'] + mx1[c']): the running maximum that makes the offspring
        // table monotone is taken on the prefix sums (lo_raw is monotone in its argument), as in k_decide
        const int perc = (nc + BS - 1) / BS;
        const int c0 = tid * perc, c1 = c0 + perc < nc ? c0 + perc : nc;
        double run = 0.0;
        // one wavefront per thread (nc <= BS): thread tid's entry is the one row_preload asked for
        const bool pre_mx = pre.have && perc <= 2;
        const double pre_vm = pre.vm, pre_vm2 = pre.vm2;
        for (int ch = c0; ch < c1; ++ch) { double vch = pipe_chunk_offset(q, ch) + (pre_mx ? (ch == c0 ? pre_vm : pre_vm2) : cmx1[ch]); run = vch > run ? vch : run; }
        double scd = wave_max_scan_d(run, lane);
        if (lane == 63) q.wredd[wave] = scd;
        __syncthreads();
        double pre = 0.0;
        for (int w = 0; w < wave; ++w) pre = q.wredd[w] > pre ? q.wredd[w] : pre;
        double before = __shfl_up(scd, 1, 64);
        if (lane > 0) pre = before > pre ? before : pre;
        run = pre;
        for (int ch = c0; ch < c1; ++ch) {
            q.pmx[ch] = run;
            double vch = pipe_chunk_offset(q, ch) + (pre_mx ? (ch == c0 ? pre_vm : pre_vm2) : cmx1[ch]);
            run = vch > run ? vch : run;
        }
        if (c1 == nc && c0 < c1) q.pmx[nc] = run;
        __syncthreads();
    }
    return d;
}
// final offspring offset from the (running-maximum) pilot prefix sum v: #{ j in [0,N) : (j+u) * S1 < N * v }  (pc.cpp:491
// scaled by N*S1: no division), guess plus exact predicate correction -- the arithmetic of k_decide's lo_at
__device__ __forceinline__ int pipe_lo_from(double v, double dn, long long Np, double S1, double invS1, double u) {
    double rhs = dn * v;
    double guess = floor(rhs * invS1 - u);
    long long g = guess < 0 ? 0 : (guess > dn ? Np : (long long)guess);
    while (g > 0 && !((((double)(g - 1)) + u) * S1 < rhs)) --g;
    while (g < Np && ((((double)g) + u) * S1 < rhs)) ++g;
    return (int)g;
}

// what the single-launch pipeline tells the extend workgroups about their row
struct PipeRow {
    int complete;          // the previous row has to be completed on load (0 for the first row of a pf_run call)
    int extend;            // 0: completion only (flush at the end of a pf_run call)
    int slot_prev;         // state ring slot to read (the general double-buffer index when !complete)
    int slot_out;          // slot to write
    double pos_prev;       // end of the previous row
    int draws;             // the draw table is kept up by this launch sequence (k_sweep): 1 + parity of the row, 0: no table
    int ahead;             // the extend role is a launch of its own and runs up to PF_RING - 2 rows ahead of the other roles
};


struct PipeLaunch {
    PipeRow row;
    int nb;                // extend workgroups of this launch (0: a launch of the bookkeeping / ledger / count roles only)
    int nblk;              // particle blocks of 256 (what the ledger's new run list is built by)
    int b_slot;            // ring slot of the row whose bookkeeping is due (-1: none)
    long long b_row;       // its row index (traces)
    double b_pos;          // its end position
    int b_set_cur;         // >= 0: the general kernels take over after this launch, with this state slot
    int lc_slot;           // ring slot of the row whose ledger upkeep and counts are due (-1: none)
    int live_slot;         // newest complete slot of the per-slot record counters (ring-overwrite check)
    int nL;                // ledger workgroups
    int ncw;               // count workgroups per epoch
    int nT;                // workgroups that keep up the draw table (k_sweep only)
    int ahead;             // as PipeRow::ahead: the rings need PF_RING entries of headroom
    int workers;           // > 0: the ledger and count work of the step is dealt out among this many workgroups (Ctrl::wq)
};


struct SweepChunk {
    KArgs A;
    long long s_begin;             // first row of this call
    long long s_last;              // last row extended by this call; steps s_last + 1 and s_last + 2 flush
    double counted_to[PF_EMAX];    // window state (CountModel::counted_to) at the start of the call
    int no_count;
    int nL_full;                   // ledger workgroups per step
    int ncw;                       // count workgroups per epoch
    int nblk;                      // particle blocks of 256
    int nT;                        // draw-table workgroups per step (0: no table)
    int split;                     // the extend role (with the draw role) and the other roles are separate launches (PF_DEBUG_SPLIT_ROLES)
    int handoff;                   // launches hand over through Ctrl::xt_done / blc_step instead of through kernel boundaries (run_sweep_flags)
    int xt_wgs;                    // workgroups of an extend / draw launch (what a slot of xt_done grows by per step)
    unsigned long long* trace;     // pf_set_wg_trace: four words per workgroup of steps [trace_t0, trace_t0 + trace_n) (k_sweep4t only)
    int trace_t0, trace_n, trace_stride;
    int workers;                   // pf_params.count_workers (PipeLaunch::workers)
};
typedef const __attribute__((address_space(4))) SweepChunk SweepChunkC;

// Waiting for another launch inside a kernel (all threads of the workgroup call it): thread 0 polls `*ctr` until it reaches `target`
// (or until somebody has reported an error: then everybody leaves at once), the workgroup barrier holds the others, and an acquire
// at agent scope -- vector caches and the scalar cache -- makes what the other launch released visible to the loads that follow.
// A workgroup that waits longer than any launch can take reports ERR_HANDOFF instead of hanging the device.
__device__ __forceinline__ bool sweep_wait_ge(Ctrl* c, const unsigned* ctr, unsigned target) {
    __shared__ int s_wait_bad;
    if (threadIdx.x == 0) {
        int bad = 0;
        long long guard = 0;
        // (relaxed polls: an acquiring load would invalidate the caches on every trip, for every workgroup that waits)
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (__hip_atomic_load(&c->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
            if (++guard > 30000000LL) { bad = 1; __hip_atomic_store(&c->err, (int)ERR_HANDOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        s_wait_bad = bad;
    }
    __syncthreads();
    const int bad_all = s_wait_bad;
    __syncthreads();                                   // (the flag may be written again by the next wait)
    return bad_all == 0;
}
// what another launch released becomes visible to the loads that follow: vector caches and the scalar cache
__device__ __forceinline__ void sweep_acquire() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_s_dcache_inv();
}
// the workgroup's stores of this launch released, its arrival counted
__device__ __forceinline__ void sweep_arrive(unsigned* ctr) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <class KA>
__device__ __forceinline__ double sweep_seg_pos(const KA& A, long long s) {       // seg_pos() of the host
    const double e = A.seg_start[s] + A.seg_len[s];
    return e < A.L ? e : A.L;
}

// what run_pipeline's `launch` lambda computes on the host, from the step's row alone
__device__ __forceinline__ bool sweep_plan(SweepChunkC& ch, long long s, int nb, PipeLaunch& PL) {
    const long long s_begin = ch.s_begin, s_last = ch.s_last;
    if (s_last < s_begin || s > s_last + 2) return false;
    const bool extend = s <= s_last, flush1 = s == s_last + 1;
    const bool have_b = extend ? (s > s_begin) : flush1;
    const bool have_lc = extend ? (s > s_begin + 1) : (flush1 ? (s_last - 1 >= s_begin) : true);
    PL.nb = nb; PL.nblk = ch.nblk;
    PL.row.extend = extend ? 1 : 0;
    PL.row.complete = ((extend && s > s_begin) || flush1) ? 1 : 0;
    PL.row.slot_prev = PL.row.complete ? (int)((s - 1) & (PF_RING - 1)) : -1;
    PL.row.slot_out = (int)(s & (PF_RING - 1));
    // only a step that completes a row needs its end; on the second flush step row s - 1 = s_last + 1 may lie past the table
    PL.row.pos_prev = (PL.row.complete && s > s_begin) ? sweep_seg_pos(ch.A, s - 1) : 0.0;
    PL.row.draws = ch.nT > 0 ? 1 + (int)(s & 1) : 0;
    PL.ahead = PL.row.ahead = (ch.split || ch.A.P > 1) ? 1 : 0;
    PL.nT = extend ? ch.nT : 0;
    PL.b_slot = have_b ? (int)((s - 1) & (PF_RING - 1)) : -1;
    PL.b_row = s - 1;
    PL.b_pos = have_b ? sweep_seg_pos(ch.A, s - 1) : 0.0;
    PL.b_set_cur = flush1 ? (int)((s_last + 1) & (PF_RING - 1)) : -1;
    PL.lc_slot = (have_lc && !ch.no_count) ? (int)((s - 2) & (PF_RING - 1)) : -1;
    PL.live_slot = (int)((s - 1) & (PF_RING - 1));
    PL.nL = PL.lc_slot >= 0 ? ch.nL_full : 0;
    PL.ncw = ch.ncw;
    PL.workers = ch.workers;
    return true;
}

