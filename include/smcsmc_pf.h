/* include/smcsmc_pf.h -- C-ABI of the MI355X (gfx950) particle-filter layer of smcsmc_amd.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json:north_star: the
 * ParticleContainer / ForestState / CountModel inner loop of the `smcsmc` binary.  The
 * reference has no FFI for this path (it is one C++ binary); the cut is made directly under
 * the public methods of the reference's ParticleContainer and CountModel, so each entry point
 * below replaces one of them (paths relative to /root/reference/src):
 *
 *   pf_create            ParticleContainer ctor arguments + PfParam/Model tables
 *                        (particleContainer.cpp:33-44, pfparam.cpp:321-380)
 *   pf_init_prior        ParticleContainer::ParticleContainer body (particleContainer.cpp:46-65)
 *   pf_load_segments     Segment buffer made resident (segdata.cpp:55-166, 182-222)
 *   pf_update_segment    ParticleContainer::update_state_to_data (particleContainer.cpp:441-466)
 *   pf_count             CountModel::extract_and_update_count (count.cpp:355-415)
 *   pf_resample          ParticleContainer::resample (particleContainer.cpp:247-311)
 *   pf_run               the do-while of pfARG_core (smcsmc.cpp:324-360) over a segment range: its rows are enqueued without
 *                        a host round trip per row (the call itself waits once, for the launches of the previous call that
 *                        still read the chunk table it is about to rewrite: bin/smcsmc calls it every 1000 rows)
 *   pf_run_many          the same loop for the chunks the front-end starts side by side, one process each
 *                        (smcsmc/model.py:1094-1098): one launch per row covers all of them
 *   pf_finish            final normalize_probability + lag-free flush (smcsmc.cpp:371-373)
 *   pf_get_counts        CountModel totals consumed by log_counts (count.cpp:66-158)
 *   pf_logl              ParticleContainer::ln_normalization_factor (particleContainer.hpp)
 *
 * Conventions: extern "C", plain pointers and sizes, caller owns host buffers, callee owns
 * device memory, one host thread per handle.  Every function returning int returns 0 on
 * success and a negative code on failure; pf_last_error() then holds the message (the
 * reference's convention is `Error: <what>` + exit 1, smcsmc.cpp:99-102).
 * There is no CPU fallback: without a HIP device pf_create fails.
 */
#ifndef SMCSMC_PF_H
#define SMCSMC_PF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pf_model {
    int32_t n_epochs;            /* E  (Model::change_times_.size()) */
    int32_t n_pops;              /* P  (1..4; scrm -I) */
    int32_t nsam;                /* haplotypes n, 2..16 */
    int32_t flags;               /* bit0 -ancestral_aware, bit1 -dephase (pfparam.cpp:143-146) */
    double loci_length;          /* Model::loci_length() */
    double mutation_rate;        /* per bp per generation */
    double recombination_rate;   /* per bp per generation */
    const double* change_times;  /* [E] generations */
    const double* pop_sizes;     /* [E*P] */
    const double* mig_rates;     /* [E*P*P] backward migration rate p -> q per generation (scrm -eM/-ema), or NULL */
    const double* single_mig;    /* [E*P*P] fixed-time moves at the start of the epoch (scrm -ej: 0 or 1), or NULL */
    const int32_t* sample_pops;  /* [nsam] or NULL */
    const int32_t* record_flags; /* [E] PfParam::record_event_in_epoch (pfparam.hpp:279-281) */
    const double* lags;          /* [E] CountModel::lags (count.cpp:230-265) */
    /* focused sampling + delayed importance weights (particle.cpp:866-891, 1020-1126; particle.hpp:59-101, 185-209);
     * n_bias_heights == 0 switches it off */
    int32_t n_bias_heights;      /* k interior band boundaries (-bias_heights, generations) */
    int32_t delay_type;          /* PfParam::ResampleDelayType: 0 recombination, 1 coalescence, 2 coal/migr; + 4 (not a reference
                                  * option): importance factors of events above the focused band are delayed like the others
                                  * instead of applied at once (particle.cpp:878-885 left out) */
    const double* bias_heights;  /* [k] */
    const double* bias_strengths;/* [k+1] */
    const double* application_delays; /* [E] Model::application_delays (smcsmc.cpp:306-307) */
    /* variational-Bayes event counts (-vb: the extra operand of -eN/-en/-eM/-ema): every coalescence / migration event
     * multiplies the particle's weights by exp_digamma(c)/c (particle.cpp:266-272); NULL = off */
    const double* vb_coal_counts;     /* [E*P] */
    const double* vb_mig_counts;      /* [E*P*P] or NULL */
    /* recombination guide (-guide; RecombinationBias, pfparam.hpp:96-223): piecewise-constant sampling rate along the
     * sequence with relative rates per sample; recombination_rate above stays the true rate.  0 segments = no guide.
     * Needs application_delays (the importance weights of guided samples are applied with delay). */
    int32_t n_rate_segments;
    int32_t reserved2;
    const double* rate_positions;     /* [K] segment starts (0-based, first = 0, no gaps) */
    const double* rate_values;        /* [K] sampling recombination rate per bp per generation */
    const double* leaf_rel_rates;     /* [K*nsam] relative rate of every sample's lineage */
} pf_model;

typedef struct pf_params {
    int64_t np;                  /* -Np */
    double ess_fraction;         /* -ESS */
    uint64_t seed;               /* -seed */
    int32_t max_trace_events;    /* resampling events whose ancestor arrays are retained for inspection */
    int32_t flags;               /* bit0: record the 100-bp local recombination map (count.cpp:559-654);
                                  * bit1: -arg, keep what pf_sample_tree_events needs;
                                  * bit2: a full delayed-factor store applies its earliest factor early instead of stopping the run
                                  *       (see delay_cap) */
    /* Capacities of the device-side rings that stand in for the reference's Arena of EvolutionaryEvents
     * (arena.cpp:56-111, unbounded there).  0 = default.  A ring that is too small for the lags in force is a
     * reported error (pf_sync returns -2, "event log ring overflow" / "generation ledger overflow"), never a
     * silent overwrite. */
    int64_t log_cap;             /* event-log records per particle slot (default 16384; with -arg 131072) */
    int64_t gen_cap;             /* resampling generations kept by the ancestor ledger (default 8192; -arg 131072) */
    int64_t piece_cap;           /* structured models: coal/migr opportunity pieces per slot (default 4*log_cap) */
    int32_t debug;               /* testing switches: PF_DEBUG_* below */
    int32_t mig_cap;             /* structured models: migration events kept per local tree (the reference's node list is
                                  * unbounded); 0 = 96.  The lists live in LDS next to the epoch tables: about 230 fit with
                                  * 32 epochs and two populations, pf_create says when a value does not.  One too many on
                                  * any tree is a reported error ("too many migration events on one local tree"). */
    int32_t count_wgs;           /* row pipeline: workgroups per epoch that share the lagged counting of a row (0 = one
                                  * per 256 particles, which is also the most).  The sums are grouped by workgroup, so the value is part
                                  * of what makes two runs bit-identical. */
    int32_t delay_cap;           /* focused sampling / guide: delayed importance factors a particle may have pending (0 = 128).  The
                                  * reference keeps them in an unbounded heap (particle.hpp:59-101, 248); here the store is a column of
                                  * delay_cap entries per particle in device memory.  One factor too many is a reported error
                                  * ("delayed-factor store overflow"), unless flags bit2 is set: then the earliest pending factor is
                                  * applied ahead of its position to make room, and pf_get_delay_stats says how often that happened. */
    int32_t count_workers;       /* row pipeline: > 0 = the ledger and count work of a step is dealt out at run time among this many workgroups
                                  * per chunk (a counter in device memory; every item still writes its own accumulators: the sums are those of
                                  * count_workers = 0, where every item is a workgroup of its own in the launch).  For several chunks per GPU
                                  * (pf_run_many), where the launch of a step is several thousand workgroups otherwise. */
    int32_t reserved4;
} pf_params;
#define PF_DEBUG_FORCE_LDS 1     /* run the LDS-tree kernels whatever nsam is */
#define PF_DEBUG_NO_FUSE   2     /* complete every row with the stand-alone k_resample (two-stream pipeline) */
#define PF_DEBUG_NO_COUNT  4     /* profiling: skip the lagged counting and the ledger upkeep */
#define PF_DEBUG_TWO_LAUNCH 8    /* rows as two launches (extend + decide) instead of the single-launch pipeline */
#define PF_DEBUG_SPLIT_ROLES 64  /* one population: the extend role and the bookkeeping / ledger / count roles of a step as two launches on
                                  * two streams, as the structured models run them (A/B against the single launch) */
#define PF_DEBUG_NO_SPEC_STAGE 32 /* row pipeline: stage the pilot scans for the parent search only once the range is known (A/B) */
#define PF_DEBUG_K_PIPE   16     /* the round-2 row paths: k_pipe (argument block passed by value, windows from the host) instead of k_sweep;
                                  * structured models: k_extend_mpr + k_decide with the counts on a second stream instead of the row pipeline */

#define PF_DEBUG_NO_DRAW_TABLE 128 /* k_sweep: every genealogy update computes its own random numbers instead of reading the ones made
                                  * ahead by the draw role (A/B; the numbers are the same) */

#define PF_DEBUG_NO_SEARCH_LUT 256 /* k_sweep: the epoch searches of an update by the four-way search instead of the bucket tables (A/B) */

#define PF_DEBUG_COUNT_YOUNG_FIRST 512 /* row pipeline: count workgroups in ascending epoch order, as before round 3 (A/B) */

#define PF_DEBUG_FLAG_HANDOFF 8192 /* one population: a row as two launches (extend + draw roles; bookkeeping + ledger + counts) that no longer wait
                                  * for each other's END: the extend launches alternate between two streams and hand the row over through
                                  * arrival counters in memory, the other launches wait the same way (run_sweep_flags; same bits) */

#define PF_DEBUG_ONE_LAUNCH (1 << 23) /* one population, at most four haplotypes, no focused sampling: every role of a step in ONE launch (k_sweep4 with the
                                  * ledger and count workgroups riding along: the form of rounds 3 and 4) instead of two -- the extend, bookkeeping
                                  * and draw roles; the ledger and count roles on the counting stream, four workgroups to a compute unit
                                  * (run_sweep_split).  A/B, same bits */

#define PF_DEBUG_CU_MASK 1024     /* with PF_DEBUG_SPLIT_ROLES: the two streams on disjoint sets of compute units (experiment) */

typedef struct pf_segments {
    int64_t n;
    const double* start;             /* [n] relative to -startpos (segdata.cpp:200-209) */
    const double* length;            /* [n] */
    const int8_t* state;             /* [n] 0 INVARIANT, 1 MISSING, 2 INVARIANT_PARTIAL */
    const int8_t* alleles;           /* [n*nsam] -1 . , 0, 1, 2 / */
    const int32_t* max_record_epoch; /* [n] max_epoch_to_update (smcsmc.cpp:266-275) */
} pf_segments;

/* Auxiliary particle filter (-apf 1..4): the per-row output of Segment::set_lookahead (segdata.cpp:225-410; the host
 * computes it, smcsmc_amd/csrc/host/segdata.cpp) and the tables of calculate_terminal_branch_length_quantiles
 * (smcsmc.cpp:128-166).  Consumed by ForestState::includeLookaheadLikelihood (particle.cpp:439-617). */
typedef struct pf_lookahead {
    int32_t level;                            /* -apf */
    int32_t max_doubletons;                   /* D: doubleton slots per row */
    int32_t n_quantiles;                      /* Q */
    int32_t reserved;
    int64_t n;                                /* rows (= segments) */
    const double* first_singleton_distance;   /* [n*nsam] */
    const double* relative_mutation_rate;     /* [n*nsam] */
    const int8_t* is_singleton_unphased;      /* [n*nsam] */
    const int32_t* n_doubletons;              /* [n] */
    const int8_t* doubleton_idx;              /* [n*D*4] seq_idx_1, seq_idx_2, unphased_1, unphased_2 */
    const double* doubleton_dist;             /* [n*D*2] first_evidence_distance, last_evidence_distance */
    const double* first_split_distance;       /* [n]  (-1: none) */
    const int8_t* split_alleles;              /* [n*nsam] allelic_state_at_first_split */
    const int32_t* split_count;               /* [n]  mutation_count_at_first_split */
    const double* quantiles;                  /* [Q] */
    const double* tbl_lengths;                /* [nsam*Q] */
    double mean_total_branch_length;
} pf_lookahead;

typedef struct pf_handle pf_handle;

/* packed count buffer, identical to the CountModel members read by count.cpp:66-158 (P == 1):
 *   coal_count[E] coal_opp[E] coal_weight[E] rec_count[E] rec_opp[E] rec_weight[E]
 *   delayed_weight_opportunity, delayed_weight_count, resample_count, ln_normalization_factor
 * raw sums without the prior pseudo-counts (count.cpp:161-227). */
#define PF_COUNTS_LEN(E) (6 * (E) + 4)
/* structured models (P > 1), CountModel members of count.hpp:95-110:
 *   coal_count[E][P] coal_opp[E][P] coal_weight[E][P]  rec_count[E] rec_opp[E] rec_weight[E]
 *   mig_count[E][P][P] mig_opp[E][P] mig_weight[E][P]  and the same four scalars */
#define PF_COUNTS_LEN2(E, P) ((P) == 1 ? PF_COUNTS_LEN(E) : (3 * (E) * (P) + 3 * (E) + (E) * (P) * (P) + 2 * (E) * (P) + 4))

const char* pf_last_error(void);
int pf_device_count(void);

pf_handle* pf_create(const pf_model* model, const pf_params* params, int device);
void pf_destroy(pf_handle* h);

int pf_init_prior(pf_handle* h, double initial_position);
/* -arg (pfparam.cpp:353-357, smcsmc.cpp:395, ParticleContainer::printTrees pc.cpp:515-555): after pf_finish, draws the
 * one particle of resample(..., NULL, 1) and returns the tree-modifying events of its history, last position first,
 * as the lines of <prefix>.trees.gz: kind 0 = R (recombination: position, height of the cut, samples under the cut
 * branch), 1 = C (coalescence of the floating lineage: position, time, samples under the node it created -- the cut
 * samples alone when it went back into its own branch).  Returns the number of events (fills at most max_events). */
int64_t pf_sample_tree_events(pf_handle* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int64_t max_events,
                              int64_t* particle_out);
/* The same with the population columns of the file (pc.cpp:541-548): from_pop = population of the event (C and M lines,
 * -1 for R), to_pop = destination of a migration (M lines, kind 2; -1 otherwise).  With several populations an update
 * yields R, C and then the migrations of the two active lineages of the walk that led to the coalescence, latest first
 * (particle.cpp:292-298; samples: those of the floating lineage, or everything else for the root's own lineage).
 * Structured models need nsam <= 8 for -arg. */
int64_t pf_sample_tree_events_pops(pf_handle* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int32_t* from_pop,
                                   int32_t* to_pop, int64_t max_events, int64_t* particle_out);

/* after pf_load_segments: switches the auxiliary particle filter on (update_lookahead_likelihood, pc.cpp:227-240) */
int pf_load_lookahead(pf_handle* h, const pf_lookahead* la);
/* calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166) over n_trees prior trees */
int pf_terminal_branch_quantiles(const pf_model* model, uint64_t seed, int64_t n_trees, const double* quantiles, int32_t nq,
                                 double* lengths_out, double* mean_total_out, int device);
int pf_load_segments(pf_handle* h, const pf_segments* segs);

/* single steps (each enqueues on the handle's stream; pf_sync waits) */
int pf_update_segment(pf_handle* h, int64_t s);
int pf_count(pf_handle* h, int64_t s, int end_data);
int pf_resample(pf_handle* h, int64_t s);
/* the hot loop over segments [s_begin, s_end): update -> count -> resample per segment */
int pf_run(pf_handle* h, int64_t s_begin, int64_t s_end);
/* the same loop for several chunks at once: rows [s_begin, s_end) of n_handles independent filters (one per chromosome
 * chunk; same device, particle count, haplotypes, epochs and options) step in lockstep through the same kernel launches, whose
 * grids cover all of them (two per row for at most four haplotypes without focused sampling -- the extend roles; the ledger and
 * count roles on the counting stream -- one otherwise).  The reference starts one process per chunk, all at once (smcsmc/model.py:1094-1098);
 * every chunk's results are bit-identical to its own pf_run.  A chunk that runs out of rows simply stops. */
int pf_run_many(pf_handle* const* handles, int32_t n_handles, int64_t s_begin, int64_t s_end);
/* 1 when pf_run_many would take these handles, 0 when the caller has to run them one after the other with pf_run (the row
 * pipeline does not apply to one of them -- several populations, more than 8 haplotypes, look-ahead, more than 131 072
 * particles -- or they differ in shape) */
int pf_can_run_many(pf_handle* const* handles, int32_t n_handles);
int pf_finish(pf_handle* h);
int pf_sync(pf_handle* h);

/* results */
int64_t pf_num_segments_done(pf_handle* h);
double pf_logl(pf_handle* h);
int pf_get_counts(pf_handle* h, double* packed, int32_t n);
int pf_get_trace(pf_handle* h, double* T, double* ess, int32_t* resampled, double* logl, int64_t n);
int pf_get_resample_events(pf_handle* h, int32_t* seg_idx, int32_t* parents, int32_t max_events);
/* structured models: the migration events on every particle's local tree ([np*cap], sorted by time; event k sits
 * on the branch above node id branch[k] and moves the lineage to newpop[k]) and the population of every
 * coalescent node ([np*(nsam-1)]); scrm keeps these as migrating unary nodes (Node::is_migrating) */
/* the local recombination map (CountModel::local_recomb_opportunity / local_recomb_counts, count.hpp:101-102):
 * opp_diff[nbins] = differential opportunity per 100-bp interval (dump_local_recomb_logs cumulates it, count.cpp:616-654),
 * counts[(nsam+2)*nbins] = per-sample, time-weighted and log-time-weighted event counts */
int pf_get_local_recomb(pf_handle* h, double* opp_diff, double* counts, int64_t nbins);
int pf_get_migrations(pf_handle* h, int32_t* n_events, double* times, int8_t* branch, int8_t* newpop, int8_t* node_pops,
                      int32_t cap);
int pf_get_particles(pf_handle* h, double* w_post, double* w_pilot, double* heights, int8_t* children,
                     double* next_base);
/* device-side timing of the kernels launched so far (HIP events on the handle's stream):
 * total milliseconds and launch count for kernel class k (0 extend, 1 decide, 2 count, 3 resample) */
int pf_get_kernel_time(pf_handle* h, int k, double* ms, int64_t* launches);
int pf_set_timing(pf_handle* h, int enable);
/* profiling builds of the library (-DPF_STAMPS) only: out == NULL enables wall-clock stamps (100 MHz ticks) of the phases of
 * the extend workgroups for the first `rows` rows; out != NULL copies them back as [rows][wavefronts][16] */
int pf_debug_stamps(pf_handle* h, int64_t rows, uint64_t* out);
/* testing (host only, no device needed): the bucket table the row kernels search an ascending table with (r_search_lut: key =
 * upper sixteen bits of the double minus *kbase, clamped to 0..255; answer = lut[key] plus one for each of the next two entries
 * that is <= t).  Returns 1 and fills lut[256] / *kbase when the table qualifies, 0 when the kernels keep the four-way search. */
int pf_test_search_lut(const double* tab, int32_t n, uint8_t* lut, int32_t* kbase);
/* measurement probe (smcsmc_amd/csrc/pf_probe.hip; not part of the filter): the hand-off of a row between `nw` wavefronts, as `rows`
 * launches back to back (mode 0: a kernel boundary per row, what the row pipeline pays) or inside one resident grid (mode 1:
 * release stores, an arrival counter per row, polling, coherent loads), with spin_ticks x 10 ns of stand-in work per wavefront
 * and row and the same reduction of all wavefronts' five partials either way.  Microseconds per row in *us_per_row. */
int pf_probe_handoff(int32_t mode, int32_t rows, int32_t nw, int64_t spin_ticks, double* us_per_row, double* checksum, int32_t device);
/* measurement aid: time stamps of every workgroup of the row kernel (k_sweep4t, pf_hip.hip; at most four haplotypes, no focused sampling)
 * for steps [first_step, first_step + n_steps) of the following pf_run / pf_run_many calls led by `h`.  pf_get_wg_trace copies four
 * 64-bit words per workgroup slot and traced step -- start, end (100 MHz clock; 0 0: the step's grid did not use the slot),
 * HW_ID | XCC_ID << 32, index within the chunk | chunk << 32 -- returns the number of words there are and fills
 * info = {steps, slots per step}.  n_steps = 0 switches the trace off.  Only the form in which every role of a step is ONE launch is
 * traced (create the handles with PF_DEBUG_ONE_LAUNCH). */
int pf_set_wg_trace(pf_handle* h, int64_t first_step, int32_t n_steps);
int64_t pf_get_wg_trace(pf_handle* h, uint64_t* out, int64_t cap_words, int32_t* info);
/* the delayed-factor store (adjustWeightsWithDelay, particle.hpp:185-209): factors applied ahead of their position because the
 * store was full (only with pf_params.flags bit2; otherwise that is an error) and the most factors any particle ever had pending */
int pf_get_delay_stats(pf_handle* h, int64_t* n_forced, int32_t* peak_pending);
/* bookkeeping for the roofline: records appended to the event log, bytes of particle state */
int pf_get_stats(pf_handle* h, int64_t* n_records, int64_t* state_bytes_per_particle, int64_t* n_resamples);

/* calculate_median_survival_distances (smcsmc.cpp:169-263): median genomic distance until an internal
 * node of a prior tree is removed, per epoch, from batches of prior ARGs simulated on the device
 * (fixed Philox seed, like the reference's MersenneTwister(true, 1)); -1-entries are filled with the
 * reference's fallbacks (smcsmc.cpp:250-258).  lags = median * lag_fraction (count.cpp:261-265). */
int pf_median_survival(const pf_model* model, uint64_t seed, int32_t min_events, int64_t max_trees, double* median_out,
                       int64_t* trees_used, int device);

/* Synthetic data next to the path (the reference shells out to scrm and converts, populationmodels.py:440-577): `nchunks`
 * independent chunks of the model's length under the same SMC' process the filter simulates (one population, or a
 * structured model with migration and joins, up to 96 migration events per local tree); per chunk
 * ascending continuous site positions pos[c*max_sites + k] and carrier masks (bit i = sample i carries the mutation).
 * n_sites[c] < 0 means more than max_sites sites were drawn (the first max_sites are returned). */
int pf_simulate_sites(const pf_model* model, uint64_t seed, int32_t nchunks, int64_t max_sites, double* pos, uint32_t* masks,
                      int64_t* n_sites, int device);

/* unit-level entry points used by the parity tests (device implementations of the math and
 * of the canonical reductions; each runs one small kernel on the handle-independent default stream) */
int pf_test_math(const double* x, int64_t n, double* out_exp, double* out_log, double* out_fastexp, int device);
int pf_test_div(const double* a, const double* b, int64_t n, double* out, int device);
int pf_test_uniform(uint64_t seed, uint32_t slot, uint32_t stream, uint64_t first_draw, int64_t n, double* out,
                    int device);
int pf_test_reduce(const double* x, int64_t n, double* out_sum, double* out_incl_scan, int device);
int pf_test_systematic(const double* pilot, int64_t n, double u, int32_t* lo, int device);

#ifdef __cplusplus
}
#endif
#endif
