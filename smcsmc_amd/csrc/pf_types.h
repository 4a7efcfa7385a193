// smcsmc_amd/csrc/pf_types.h -- structures shared by the host code and all kernels of the particle filter
// (device-visible state, per-step control block, kernel argument block, event-log record helpers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define PF_EMAX 64
#define PF_STAMP_W 32        // profiling builds: words per wavefront and row in KArgs::stamps
#define PF_DCAP_DEFAULT 128   // pending delayed factors per particle unless pf_params.delay_cap says otherwise (the reference's heap,
                              // particle.hpp:248, is unbounded; a full store is a reported error)
#define PF_BIAS_MAX 8         // interior bias heights
#define PF_DECIDE_TAB 16384   // offspring / parent tables fit LDS (16-bit entries) up to this many particles
#define PF_RING 16             // slots of the row pipeline's rings (state, scans, partials, per-row control data): row s lives in slot s & 15.
                               // Four would do for one launch per row (the counts of row s - 2 ride in launch s); sixteen let the extend
                               // launches of the two-launch form run up to eight steps between two waits for the counting stream
#define PF_LEDGER_BLOCKS 192   // extra workgroups of k_resample that maintain the ancestor ledger
#define REC_RECOMB 1
#define REC_COALMIGR 2

// ------------------------------------------------------------------ device-visible structures
struct DState {
    double* S;        // [(n-1)][Np]
    int8_t* C;        // [2(n-1)][Np]
    double* w_post;   // [Np]
    double* w_pilot;
    double* next_base;
    double* x_mark;
    double* Ltree;
    int* mark_limit;
    // delayed importance factors (particle.hpp:59-101, 185-209); allocated only with focused sampling
    double* total_delayed;   // [Np]
    int* dcount;             // [Np]
    double* dpos;            // [dcap][Np] application positions
    double* dfac;            // [dcap][Np]
    double* ddelta;          // [dcap][Np]
    int* dk;                 // [dcap][Np]
    int* ridx;               // [Np] guide segment the particle is in (_current_seq_idx); allocated with a guide
    double* lookahead;       // [Np] lookahead_weight_ (auxiliary particle filter); allocated with pf_load_lookahead
    // structured models (pf_mp.h); allocated only when P > 1
    int8_t* Pn;              // [(n-1)][Np] population of every coalescent node
    int* nm;                 // [Np] migration events on the local tree
    double* Mt;              // [mcap][Np]
    int8_t* Mb;              // [mcap][Np]
    int8_t* Mq;              // [mcap][Np]
};

struct Ctrl {
    double cur_pos;        // site_where_weight_was_updated_ (same for every particle)
    double logl;           // ln_normalization_factor_
    double inv_T, T, S1, S2, ess, u;
    double delayed_opp;
    double delayed_count;
    double counted_to[PF_EMAX];
    double update_to[PF_EMAX];
    long long n_resample;
    int flag;              // resample at this segment?
    int cur;               // index of the live state buffer
    int gen;               // current generation (number of resampling events so far)
    int first_epoch;       // first epoch updated by the current count step (E = none)
    int g_retain;          // oldest generation whose run list is still maintained
    int g_safe;            // g_retain as of two rows before the last one booked: no count that is running or still to run asks for an older
                           // generation (what an extend role that runs ahead of the counts checks its log writes against)
    int g_lo[PF_EMAX];     // generation containing counted_to[e]
    int g_hi[PF_EMAX];     // generation containing update_to[e] of the current count step
    int err;               // sticky error code
    int count_active;
    int end_seq;
    int pending_fin;       // k_count partials of the previous step still have to be folded into the totals
    int delay_peak;        // most delayed factors any particle ever had pending
    unsigned long long n_delay_evict;   // factors applied ahead of their position because the store was full (delay_evict only)
    long long nres_prev;   // n_resample as of the end of the last k_resample (stable during k_decide)
    // what the counting stream needs to know about a step, double-buffered by step parity
    struct StepInfo { double inv_T; int G; int flag; } step[2];
    int gen_prev;          // generation index as of the end of the last k_resample (stable during k_decide)
    int nbx_used;
    int lver;              // which of the two copies of the run lists is in force (the single-launch pipeline writes the
                           // re-based lists into the other copy; every other kernel updates them in place)
    // Single-launch row pipeline (k_pipe): everything a later launch needs to know about row r, in slot r & (PF_RING - 1).
    // Written by the bookkeeping workgroup of launch r + 1, read by launches r + 2 and r + 3.
    struct RowInfo {
        double T, inv_T, S1, u, pos;
        long long n_res;       // resampling events before this row's decision (= index of its uniform)
        int gen;               // generation the row's particles belong to
        int flag;              // the row ended with a resampling
        int g_retain, first;
        int lver;              // run-list copy in force for this row's counts (before its own re-basing)
        int pad;
        double wa[PF_EMAX], wb[PF_EMAX];   // count windows of the row (count.cpp:363-385)
        int g_lo[PF_EMAX], g_hi[PF_EMAX];
    } ri[PF_RING];
    // what the extend role itself notes about the row it completes (slot r & (PF_RING - 1)): the structured models' extend launches run
    // on their own stream and do not wait for the bookkeeping role, which derives the same numbers on the counting stream
    struct ExtendNote { long long n_res; int gen; int flag; } xr[PF_RING];
    // Hand-off between launches through memory (PF_DEBUG_FLAG_HANDOFF, run_sweep_flags): the extend / draw launch of step t is enqueued
    // without waiting for the launch of step t - 1 to end; its workgroups wait here for the arrivals of that launch's workgroups.
    // Counters only grow during a pf_run call (slot t & (PF_RING - 1) is at (t / PF_RING + 1) x workgroups-per-launch when step t is done).
    unsigned xt_done[PF_RING];   // arrivals of the extend / draw launches
    unsigned blc_arrive;         // arrivals of the bookkeeping / ledger / count launch in flight (they run one after the other)
    int blc_step;                // steps whose bookkeeping / ledger / count launch has ended
    unsigned wq[PF_RING];        // pf_params.count_workers: next ledger / count work item of the row in that ring slot (zeroed by the bookkeeping of
                                 // the step before the one whose workers use it)
    double last1[PF_RING];       // pilot scan value at the last particle of the row in ring slot k (= oracle incl[N-1] minus chunk offset)
};

struct KArgs {
    // model
    int E, n, flags;
    double L, mu, rho;
    const double* T;
    int rec_trees;                 // -arg: records carry the descendants of the new node, rings hold the whole chunk
    const double* inv2N;
    const double* Hc;              // [E] cumulative coalescence intensity at the epoch starts: Hc[e+1] = Hc[e] + (T[e+1]-T[e]) * inv2N[e]
    const double* lags;
    const int* recflags;
    // structured models: P populations
    int P, ncol;                   // ncol = statistics per epoch (6 when P == 1)
    const double* inv2Np;          // [E*P]
    const double* mig_rate;        // [E*P*P]
    const double* mig_tot;         // [E*P]
    const int* join_map;           // [E*P]
    const double* cum_coal;        // [E*P] cumulative coalescence intensity per population at the epoch starts
    const double* cum_mig;         // [E*P] cumulative emigration intensity per population at the epoch starts
    const double* next_join;       // [E]   start of the next epoch with a fixed-time population move
    const int* next_join_epoch;    // [E]   that epoch (E if none)
    int mcap;                      // migration events kept per local tree (pf_params.mig_cap, default PF_MMAX)
    const int* sample_pop;         // [n]
    double* plog;                  // coal/migr opportunity pieces: plog[(p*pcap + k%pcap)*3 .. +3)
    unsigned pcap;
    unsigned* pidx;                // slot-owned: pieces ever written by this slot
    // auxiliary particle filter (pf_lookahead)
    int apf, la_D, la_Q;
    const double* la_fsd; const double* la_rmr; const int8_t* la_unph; const int* la_nd; const int8_t* la_didx;
    const double* la_ddist; const double* la_split; const int8_t* la_salleles; const int* la_sk;
    const double* la_q; const double* la_tbl;
    double la_mean_tbl;
    // local recombination map (count.cpp:559-654): differential opportunity [lmap_bins] and per-sample / time /
    // log-time weighted counts [(n+2)][lmap_bins] per 100-bp interval; null when not recorded
    double* lmap_opp;
    double* lmap_cnt;
    long long lmap_bins;
    // variational-Bayes weight factors exp_digamma(c)/c (particle.cpp:266-272), computed on the device from the event
    // counts so that they carry the device's exp / log; null = off
    const double* vb_coal;         // [E*P]
    const double* vb_mig;          // [E*P*P]
    // recombination guide (RecombinationBias, pfparam.hpp:96-223): g_K segments (0 = none)
    int g_K;
    const double* g_pos; const double* g_rho; const double* g_leaf;
    // run parameters
    long long Np;
    double ess_threshold;
    unsigned long long seed;
    // focused sampling
    int n_bias, delay_type;
    int dcap, delay_evict;         // capacity of the delayed-factor store per particle; full store: apply the earliest factor early (and count) instead of stopping
    double bias_H[PF_BIAS_MAX + 2];
    double bias_S[PF_BIAS_MAX + 1];
    const double* app_delays;
    int* chunk_dpend;              // [nc] particles with pending delayed factors, per wavefront
    double delayed_count_unused;
    // state: every array of DState holds `nslots` copies back to back and st0 points at copy 0 (state_slot() below gives
    // copy k).  Copies 0 and 1 are the double buffer of the general kernels; the single-launch pipeline uses four as a
    // ring indexed by row & (PF_RING - 1) (a row's raw weights, tree and stretch stay readable for the counts two launches later).
    // Pointer arithmetic instead of an array of structs: a kernel argument indexed at run time would be copied to the
    // stack.
    DState st0;
    int nslots;
    unsigned long long* rng_ctr;   // slot-owned
    // draw table of the row pipeline (draw_role): [Np][PF_DRAW_RING] entries of two doubles, the first block index that is not in
    // it and the counter at the start of a row, both [2][Np] by row parity (one launch writes what the next one reads)
    double* dt_tab; unsigned long long* dt_filled; unsigned long long* dt_ctr;
    // bucket tables of the epoch search and of the inverse of the cumulative intensity (r_search_lut): 2 x 256 bytes, the
    // key of bucket 0 of either; null when a table does not qualify (then the four-way search runs)
    const unsigned char* lut; int lut_kbT, lut_kbH;
    double* ebuf;                  // slot-owned
    unsigned* widx;                // slot-owned: records ever appended by this slot
    // event log: rec[(p*cap + k%cap)*RS .. +RS)
    double* log;
    unsigned cap;
    int RS;
    // ancestor ledger (rings of Gcap generations)
    int Gcap;
    unsigned* gstart;              // [Gcap][Np]  widx at the start of generation g
    int* lo;                       // [Gcap][Np+1] offspring ranges of resampling event r (between gen r and r+1)
    double* gen_x0;                // [Gcap] position where generation g starts
    int* parent;                   // [Np] parent slot of every new slot at the current resampling event
    int* blkcnt2[2];               // [nblocks] survivors per particle workgroup at the resampling event of a step (by step parity)
    // run-length encoded composite ancestor maps: generation g's list maps the slots of the
    // current generation to slots of generation g: run i covers [run_st[i], run_st[i+1]) -> run_anc[i]
    int* run_st;                   // [Gcap][Np]
    int* run_anc;                  // [Gcap][Np]
    int* nruns;                    // [Gcap]
    int* run_st2;                  // second copy of the three (Ctrl::lver), allocated for the single-launch pipeline
    int* run_anc2;
    int* nruns2;
    // rings of the single-launch pipeline, slot = row & (PF_RING - 1): per-particle scans [PF_RING][Np], per-wavefront partials [PF_RING][nc],
    // survivors per workgroup [PF_RING][nblocks], records appended per slot [PF_RING][Np]
    double* rg_scan1; double* rg_scan1m; double* rg_scanp;
    double* rg_cpost; double* rg_csq; double* rg_cpil; double* rg_cpp; double* rg_cmx1; double* rg_coffp;
    int* rg_dpend; int* rg_blkcnt;
    int blk_gran;                  // entries of rg_blkcnt per block of 256 particles: 1, or 4 when the extend workgroups own 64 particles
    unsigned* rg_widx;
    int nc;                        // wavefronts = (Np + 63) / 64
    // profiling builds (-DPF_STAMPS): wall-clock stamps of the extend workgroups' phases, [rows][nc][PF_STAMP_W]; null otherwise
    unsigned long long* stamps;
    long long stamp_rows;
    // per-wavefront partials written by k_extend
    double* chunk_post;            // [nc]
    double* chunk_sq;
    double* chunk_pil;
    double* scan1;                 // [Np] within-wavefront inclusive scan of the pilot weights
    double* chunk_off;             // [nc]
    double* l2scan;                // [nc]
    double* scanp2[2];             // [Np] within-wavefront inclusive scan of the posterior weights (by step parity)
    // snapshot of what k_count needs from the live particles of a step (by step parity): the counting stream
    // runs concurrently with k_resample / the next k_extend, which rewrite the live state
    double* snap_w[2];             // [Np] raw posterior weight
    double* snap_S[2];             // [(n-1)][Np]
    double* snap_xm[2];            // [Np] x_mark
    int* snap_ml[2];               // [Np] mark_limit
    unsigned* snap_widx[2];        // [Np]
    int sp;                        // parity of the step this launch belongs to (set by the host per launch)
    double* scan1m;                // [Np] running max of scan1 inside the wavefront
    double* chunk_mx1;             // [nc] max of scan1 per wavefront
    double* chunk_pp;              // [nc] its per-wavefront totals
    double* chunk_offp2[2];        // [nc] exclusive offsets of the posterior scan (by step parity)
    double* l2scanp;               // [nc]
    // counting
    double* totals;                // [6][E]
    double* partial;               // [E][nbx][6]
    int nbx;
    const int* cw_off;             // row pipeline: [E + 1] first count workgroup of the j-th column, columns in the order oldest epoch first
                                   // (the old epochs' windows hold nearly every particle, the young ones' a few ancestors: their columns
                                   // are given fewer workgroups); null: PipeLaunch::ncw workgroups for every column
    // segments
    const double* seg_start;
    const double* seg_len;
    const int8_t* seg_state;
    const int8_t* seg_alleles;
    const int* seg_limit;
    // traces
    double* tr_T;
    double* tr_ess;
    double* tr_logl;
    int* tr_flag;
    int* ev_seg;
    int* ev_parents;
    int max_trace_events;
    Ctrl* ctrl;
};

// copy k of the particle state (see KArgs::st0); fields that are not allocated stay unusable, as in st0
template <class KA>
__host__ __device__ inline DState state_slot(const KA& A, int k) {
    // member by member: A may live in the constant address space (KArgsC), from which a struct cannot be copy-constructed
    DState d;
    d.S = A.st0.S; d.C = A.st0.C; d.w_post = A.st0.w_post; d.w_pilot = A.st0.w_pilot; d.next_base = A.st0.next_base;
    d.x_mark = A.st0.x_mark; d.Ltree = A.st0.Ltree; d.mark_limit = A.st0.mark_limit;
    d.total_delayed = A.st0.total_delayed; d.dcount = A.st0.dcount; d.dpos = A.st0.dpos; d.dfac = A.st0.dfac;
    d.ddelta = A.st0.ddelta; d.dk = A.st0.dk; d.ridx = A.st0.ridx; d.lookahead = A.st0.lookahead;
    d.Pn = A.st0.Pn; d.nm = A.st0.nm; d.Mt = A.st0.Mt; d.Mb = A.st0.Mb; d.Mq = A.st0.Mq;
    const size_t K = (size_t)k, Np = (size_t)A.Np, n1 = (size_t)(A.n - 1);
    d.S += K * n1 * Np; d.C += K * 2 * n1 * Np;
    d.w_post += K * Np; d.w_pilot += K * Np; d.next_base += K * Np; d.x_mark += K * Np; d.Ltree += K * Np; d.mark_limit += K * Np;
    d.total_delayed += K * Np; d.dcount += K * Np;
    d.dpos += K * (size_t)A.dcap * Np; d.dfac += K * (size_t)A.dcap * Np; d.ddelta += K * (size_t)A.dcap * Np; d.dk += K * (size_t)A.dcap * Np;
    d.ridx += K * Np; d.lookahead += K * Np;
    d.Pn += K * n1 * Np; d.nm += K * Np;
    d.Mt += K * (size_t)A.mcap * Np; d.Mb += K * (size_t)A.mcap * Np; d.Mq += K * (size_t)A.mcap * Np;
    return d;
}

// The argument block as the kernels of the multi-chunk sweep see it: one KArgs per chunk in device memory, read through
// the constant address space so that every field is a scalar load issued where it is used (a kernel gets a plain
// pointer and casts it).  Device functions take the block as `const KA&` and work on either form.
typedef const __attribute__((address_space(4))) KArgs KArgsC;

// Count windows of one step (count.cpp:363-385).  The rule depends only on segment positions and lags,
// so the host evaluates it (host_first_epoch) and hands the result to the kernels by value.
struct Windows {
    int first;                 // first epoch that updates (E = none)
    int end_data;
    double a[PF_EMAX];         // counted_to before this step
    double b[PF_EMAX];         // update_to of this step
};

enum { ERR_LOG_OVERFLOW = 1, ERR_GEN_OVERFLOW = 2, ERR_ZERO_PROB = 3, ERR_COUNT_MISMATCH = 4, ERR_MIG_OVERFLOW = 5,
       ERR_MP_INTERNAL = 6, ERR_NO_COALESCENCE = 7, ERR_DELAY_OVERFLOW = 8, ERR_HANDOFF = 9 };

// record meta word: type | lim_start+1 << 8 | lim_event+1 << 16 | n_eff << 24 | descendants << 32 (samples below the
// branch cut by the recombination that ends the stretch, bit i = sample i; descendants.hpp:22-33) | descendants of the
// node the floating lineage created << 48 (what the C line of a .trees.gz carries; only the cut samples themselves
// when the lineage went back into its own branch)
__device__ __forceinline__ unsigned long long make_meta(int type, int lim_start, int lim_event, int n_eff, unsigned desc = 0,
                                                        unsigned desc_new = 0) {
    return (unsigned long long)(type & 0xff) | ((unsigned long long)((lim_start + 1) & 0xff) << 8) |
           ((unsigned long long)((lim_event + 1) & 0xff) << 16) | ((unsigned long long)(n_eff & 0xff) << 24) |
           ((unsigned long long)(desc & 0xffff) << 32) | ((unsigned long long)(desc_new & 0xffff) << 48);
}

template <class KA>
__device__ __forceinline__ double* rec_ptr(const KA& A, long long p, unsigned k) {
    return A.log + ((size_t)p * A.cap + (k % A.cap)) * A.RS;
}

