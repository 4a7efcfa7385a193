/* oracle/smc_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * C API of the CPU oracle: a single-threaded restatement of the smcsmc
 * particle-filter forward sweep (reference: /root/reference/src/particleContainer.cpp,
 * particle.cpp, count.cpp, smcsmc.cpp:278-401).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY STATUS: "parity unpinned" against the upstream binary -- the reference's
 * coalescent engine (scrm fork) is an un-vendored submodule and Boost is absent, so the
 * reference cannot be built here, and its own tests hold no golden vectors for weights,
 * resampling indices or log-likelihood.  The oracle is pinned
 * (a) structurally, function by function against the cited reference lines,
 * (b) by the reference's .seg fixtures / FormatDouble contract for the boundary, and
 * (c) distributionally (coalescent prior expectations, known simulation truth), and
 * (d) by the acceptance bands the reference's own regression tests hold (tests/golden/reference_bands.json,
 *     transcribed from test/old/newtests by tests/golden/make_reference_bands.py): the no-data classes are
 *     asserted on this oracle in tests/test_oracle_cpu.py, all of them through the GPU path in
 *     tests/test_gpu_reference_bands.py; DESIGN.md section 6 has the pass/fail table.
 *
 * The struct layouts below are deliberately identical to include/smcsmc_pf.h so the
 * parity tests can feed both sides the same buffers.
 */
#ifndef SMC_ORACLE_H
#define SMC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smco_model {
    int32_t n_epochs;            /* E */
    int32_t n_pops;              /* P (1..8) */
    int32_t nsam;                /* number of haplotypes n (2..16) */
    int32_t flags;               /* bit0 ancestral_aware, bit1 dephase */
    double loci_length;          /* L, bp */
    double mutation_rate;        /* per bp per generation */
    double recombination_rate;   /* per bp per generation */
    const double* change_times;  /* [E] generations, change_times[0] == 0 */
    const double* pop_sizes;     /* [E*P] diploid N_e per epoch */
    const double* mig_rates;     /* [E*P*P] per generation, or NULL */
    const double* single_mig;    /* [E*P*P] -ej style probabilities, or NULL */
    const int32_t* sample_pops;  /* [nsam] or NULL */
    const int32_t* record_flags; /* [E] bit0 record recomb, bit1 record coal/migr (pfparam.hpp:279-281) */
    const double* lags;          /* [E] bp (count.cpp:230-265) */
    /* focused sampling + delayed importance weights (optional; n_bias_heights == 0 switches it off) */
    int32_t n_bias_heights;      /* k interior band boundaries (-bias_heights, generations) */
    int32_t delay_type;          /* 0 recombination height, 1 first coalescence, 2 first coal/migr (pfparam.hpp) */
    const double* bias_heights;  /* [k] */
    const double* bias_strengths;/* [k+1] */
    const double* application_delays; /* [E] bp (smcsmc.cpp:306-307) */
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
} smco_model;

typedef struct smco_params {
    int64_t np;                  /* number of particles */
    double ess_fraction;         /* -ESS (pfparam.cpp:323) */
    uint64_t seed;
    int32_t max_trace_events;    /* number of resampling events whose ancestor arrays are kept */
    int32_t mig_cap;             /* migration events a local tree may hold before the run stops (0 = 96; at most 256):
                                  * the capacity of the device path, pf_params.mig_cap */
    int32_t delay_cap;           /* pending delayed importance factors a particle may hold (0 = 128, the device path's default,
                                  * pf_params.delay_cap).  The reference's heap is unbounded (particle.hpp:248); here a full store
                                  * is a reported error ("delayed-factor store overflow") ... */
    int32_t delay_evict;         /* ... unless this is non-zero: then the earliest pending factor is applied ahead of its
                                  * position to make room, and smco_get_delay_stats counts how often */
} smco_params;

typedef struct smco_segments {
    int64_t n;
    const double* start;             /* [n] 0-based, relative to -startpos (segdata.cpp:200-209) */
    const double* length;            /* [n] */
    const int8_t* state;             /* [n] 0 INVARIANT, 1 MISSING, 2 INVARIANT_PARTIAL (segdata.hpp:84) */
    const int8_t* alleles;           /* [n*nsam] -1 missing, 0, 1, 2 unphased het */
    const int32_t* max_record_epoch; /* [n] smcsmc.cpp:266-275 */
} smco_segments;

/* auxiliary particle filter (-apf 1..4): the per-row output of Segment::set_lookahead (segdata.cpp:225-410) and the
 * tables of calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166); same layout as pf_lookahead */
typedef struct smco_lookahead {
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
} smco_lookahead;

/* layout of the packed count buffer (doubles), P == 1:
 *   [0*E..1*E) coal_count  [1*E..2*E) coal_opp  [2*E..3*E) coal_weight
 *   [3*E..4*E) rec_count   [4*E..5*E) rec_opp   [5*E..6*E) rec_weight
 *   [6*E+0] delayed_weight_opportunity [6*E+1] delayed_weight_count
 *   [6*E+2] resample_count [6*E+3] ln_normalization_factor
 * (raw sums, WITHOUT the prior pseudo-counts of count.cpp:161-227; the host adds those.) */
#define SMCO_COUNTS_LEN(E) (6 * (E) + 4)
/* P > 1:  coal_count[E][P] coal_opp[E][P] coal_weight[E][P]  rec_count[E] rec_opp[E] rec_weight[E]
 *         mig_count[E][P][P] mig_opp[E][P] mig_weight[E][P]  and the same four scalars (count.hpp:95-110) */
#define SMCO_COUNTS_LEN2(E, P) ((P) == 1 ? SMCO_COUNTS_LEN(E) : (3 * (E) * (P) + 3 * (E) + (E) * (P) * (P) + 2 * (E) * (P) + 4))

void* smco_create(const smco_model* m, const smco_params* p);
void smco_destroy(void* h);
const char* smco_last_error(void);

int smco_load_lookahead(void* h, const smco_lookahead* la);            /* switches the auxiliary particle filter on */
/* calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166) over n_trees prior trees (Philox stream 3) */
int smco_terminal_branch_quantiles(const smco_model* m, uint64_t seed, int64_t n_trees, const double* quantiles, int32_t nq,
                                   double* lengths_out, double* mean_total_out);
int smco_init_prior(void* h, double initial_position);               /* particleContainer.cpp:33-65 */
int smco_run(void* h, const smco_segments* segs);                    /* smcsmc.cpp:324-373 */
/* single steps, mirroring the reference's public methods */
int smco_update_segment(void* h, const smco_segments* segs, int64_t s);  /* pc.cpp:441-466 */
int smco_count(void* h, double current_base, int end_data);             /* count.cpp:355-415 */
int smco_resample(void* h, double update_to);                           /* pc.cpp:247-311; returns 1 if resampled */
int smco_finish(void* h);                                                /* smcsmc.cpp:371-373 */

int64_t smco_num_segments_done(void* h);
int smco_get_trace(void* h, double* T, double* ess, int32_t* resampled, double* logl, int64_t n);
int smco_get_resample_events(void* h, int32_t* seg_idx, int32_t* parents, int32_t max_events);
int smco_get_particles(void* h, double* w_post, double* w_pilot, double* heights, int8_t* children,
                       double* next_base);
int smco_get_counts(void* h, double* packed, int32_t n);
/* 100-bp local recombination map (count.cpp:559-654): call enable before init_prior; opp_diff[nbins] is the
 * differential opportunity of count.hpp:101, counts[(nsam+2)*nbins] the per-sample, time and log-time weighted counts */
int smco_enable_local_recomb(void* h);
int smco_enable_tree_recording(void* h);
int64_t smco_sample_tree_events(void* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int64_t max_events,
                                int64_t* particle_out);
int64_t smco_sample_tree_events_pops(void* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int32_t* from_pop,
                                     int32_t* to_pop, int64_t max_events, int64_t* particle_out);
int smco_get_local_recomb(void* h, double* opp_diff, double* counts, int64_t nbins);
/* structured models: migration events kept on each particle's local tree ([np*cap], sorted by time) and the
 * population of every coalescent node ([np*(nsam-1)]) */
int smco_get_migrations(void* h, int32_t* n_events, double* times, int8_t* branch, int8_t* newpop, int8_t* node_pops,
                        int32_t cap);
double smco_logl(void* h);
/* work statistics for DESIGN.md / roofline bookkeeping */
int smco_get_stats(void* h, int64_t* n_recombinations, int64_t* n_events_allocated, int64_t* n_resamples);

/* delayed-factor store: factors applied early to make room (delay_evict), most factors any particle had pending */
int smco_get_delay_stats(void* h, int64_t* n_forced, int32_t* peak_pending);

/* calculate_median_survival_distances (smcsmc.cpp:169-263), batched like the HIP driver */
int smco_median_survival(const smco_model* m, uint64_t seed, int32_t min_events, int64_t max_trees, double* median_out,
                         int64_t* trees_used);

/* exposed helpers so the tests can pin the math bit-for-bit against the HIP side */
double smco_exp(double x);
double smco_log(double x);
double smco_fastexp(double x);
double smco_uniform(uint64_t seed, uint32_t slot, uint32_t stream, uint64_t draw);
/* canonical reductions (DESIGN.md "canonical arithmetic") */
double smco_canon_sum(const double* x, int64_t n);
void smco_canon_scan(const double* x, double* incl, int64_t n);
/* systematic resampling on given weights (pc.cpp:474-504): returns offspring start indices lo[0..n] */
void smco_systematic(const double* pilot, int64_t n, double u, int32_t* lo);

#ifdef __cplusplus
}
#endif
#endif
