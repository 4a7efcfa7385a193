/* oracle/smc_oracle.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Single-threaded restatement of the smcsmc particle-filter forward sweep.  Each function
 * cites the reference lines it follows (paths relative to /root/reference/src).
 * PARITY STATUS: "parity unpinned" against the upstream binary (see smc_oracle.h).
 *
 * What is restated verbatim (arithmetic order included):
 *   extend_ARG                 particle.cpp:743-918
 *   calculate_likelihood       particle.cpp:625-680
 *   trackLocalTreeBranchLength particle.cpp:699-730
 *   update_weight_at_site      particleContainer.cpp:138-224
 *   normalize_probability      particleContainer.cpp:420-438
 *   resample / systematic      particleContainer.cpp:247-311, 474-504, 321-392
 *   event records + counting   particle.cpp:193-390, coalevent.hpp:209-244, count.cpp:355-555
 * What is reconstructed (the scrm fork is absent): the SMC' genealogy
 * update (Forest::sampleNextGenealogy / sampleCoalescences, mirrored by particle.cpp:1266-1384):
 * cut the local tree at a uniform point, let the floating lineage coalesce upwards at rate
 * (#contemporaries)/(2N(t)) through time intervals delimited by node heights and epoch
 * boundaries, with the buffered unit exponential of RandomGenerator::sampleExpoLimit.
 *
 * Documented representation choices (DESIGN.md "Deviations"):
 *   D1 counter-based Philox stream per particle slot instead of one shared MersenneTwister
 *   D2 multiplicities expanded: always Np records of multiplicity 1 (particle.cpp:831-857)
 *   D3 canonical radix-64 reduction / scan order for sums (exact definition in canon_* below)
 *   D4 u_j = (j+U)/N in closed form instead of the running sum of particleContainer.cpp:494
 *   D5 ESS computed as S1*S1/S2 on the un-normalised pilot weights (scale invariant)
 *   D6 recombination-opportunity rectangles are closed when the stretch ends instead of being
 *      written with the sampled next_base and patched on resampling (particle.cpp:393-436);
 *      the sums of count.cpp:495-555 are identical
 *   D12 one population: the floating lineage's coalescence time is read off the cumulative intensity
 *      Hc(t) = int_0^t ds/(2N(s)) (one comparison per node passed, then the inverse) instead of interval by
 *      interval; same distribution, and the arithmetic the device uses.  One opportunity record per interval is
 *      still written, between the cut height and the time found.
 *   D13 one population: the four uniforms of a genealogy update are the halves of two consecutive Philox blocks
 *   D14 structured models: the walk of the floating lineage compares its budget with differences of per-population
 *      cumulative intensities once per stretch between two changes of the configuration (node, migration event on
 *      the tree, fixed-time move) instead of once per epoch; same distribution (mp_coalesce below)
 *   (D7-D11, R1, R2: DESIGN.md section 6)
 */
#include "smc_oracle.h"
#include "smc_math.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

namespace smco {

static thread_local std::string g_err;

enum { NMAX = 16 };
enum { MMAX = 256 };  /* storage for migration events per local tree (multi-population models); the limit in force is
                       * Filter::mig_cap (smco_params.mig_cap, default 96, as pf_params.mig_cap of the device path) */
enum { DCAP_DEFAULT = 128 };   /* default capacity of the per-particle delayed-factor store: the capacity of the device path
                               * (pf_params.delay_cap).  The reference's heap is unbounded (particle.hpp:248); a full store is a
                               * reported error unless smco_params.delay_evict asks for the oldest factor to be applied early */
enum { REC_RECOMB = 1, REC_COALMIGR = 2 };

/* ------------------------------------------------------------------ canonical reductions */

/* pairwise tree over 64 slots: identical association to a 64-lane xor-butterfly */
static double tree64(const double* x, int64_t n) {
    double v[64];
    for (int i = 0; i < 64; ++i) v[i] = i < n ? x[i] : 0.0;
    for (int m = 1; m < 64; m <<= 1)
        for (int i = 0; i < 64; i += 2 * m) v[i] = v[i] + v[i + m];
    return v[0];
}

/* always three radix-64 levels (supports n <= 262144) */
static double canon_sum(const double* x, int64_t n) {
    if (n > 262144) throw std::runtime_error("canon_sum: n too large");
    std::vector<double> l1((n + 63) / 64);
    for (int64_t c = 0; c < (int64_t)l1.size(); ++c) l1[c] = tree64(x + c * 64, std::min<int64_t>(64, n - c * 64));
    std::vector<double> l2((l1.size() + 63) / 64);
    for (int64_t c = 0; c < (int64_t)l2.size(); ++c)
        l2[c] = tree64(l1.data() + c * 64, std::min<int64_t>(64, (int64_t)l1.size() - c * 64));
    return tree64(l2.data(), (int64_t)l2.size());
}

/* Hillis-Steele inclusive scan of up to 64 values (the association a wave scan produces) */
static void hs64(const double* x, double* out, int64_t n) {
    double v[64], w[64];
    for (int i = 0; i < 64; ++i) v[i] = i < n ? x[i] : 0.0;
    for (int d = 1; d < 64; d <<= 1) {
        for (int i = 0; i < 64; ++i) w[i] = i >= d ? v[i - d] + v[i] : v[i];
        for (int i = 0; i < 64; ++i) v[i] = w[i];
    }
    for (int i = 0; i < n; ++i) out[i] = v[i];
}

static void canon_scan(const double* x, double* incl, int64_t n) {
    if (n > 262144) throw std::runtime_error("canon_scan: n too large");
    int64_t nc = (n + 63) / 64;
    std::vector<double> l1(n), tot(nc), l2(nc);
    for (int64_t c = 0; c < nc; ++c) {
        int64_t m = std::min<int64_t>(64, n - c * 64);
        hs64(x + c * 64, l1.data() + c * 64, m);
        tot[c] = l1[c * 64 + m - 1];
    }
    int64_t ng = (nc + 63) / 64;
    std::vector<double> base(ng);
    double run = 0.0;
    for (int64_t g = 0; g < ng; ++g) {
        int64_t m = std::min<int64_t>(64, nc - g * 64);
        hs64(tot.data() + g * 64, l2.data() + g * 64, m);
        base[g] = run;
        run = run + l2[g * 64 + m - 1];
    }
    for (int64_t c = 0; c < nc; ++c) {
        double off = (c % 64 == 0) ? 0.0 : l2[c - 1];
        double chunk_off = base[c / 64] + off;
        int64_t m = std::min<int64_t>(64, n - c * 64);
        for (int64_t i = 0; i < m; ++i) incl[c * 64 + i] = chunk_off + l1[c * 64 + i];
    }
}

/* particleContainer.cpp:474-504 in closed form (D4): sample j has quantile u_j = (j+U)/N and goes to the first
 * record whose upper partial sum exceeds it (pc.cpp:491: partial_sum[i+1]/total > u_j).  With both sides
 * multiplied by N*total:  lo[i] = #{ j in [0,N) : (j+U) * total < N * incl[i-1] }  (no division). */
static void systematic(const double* incl, int64_t n, double u, int32_t* lo) {
    const double total = incl[n - 1];
    const double dn = (double)n;
    const double inv_total = 1.0 / total;
    lo[0] = 0;
    for (int64_t i = 1; i < n; ++i) {
        double rhs = dn * incl[i - 1];
        double guess = std::floor(rhs * inv_total - u);
        int64_t g = guess < 0 ? 0 : (guess > dn ? n : (int64_t)guess);
        while (g > 0 && !((((double)(g - 1)) + u) * total < rhs)) --g;
        while (g < n && ((((double)g) + u) * total < rhs)) ++g;
        lo[i] = (int32_t)g;
    }
    lo[n] = (int32_t)n;
    /* the scan is not guaranteed monotone in floating point: enforce non-decreasing offsets */
    for (int64_t i = 1; i <= n; ++i) lo[i] = std::max(lo[i], lo[i - 1]);
}

/* ------------------------------------------------------------------ model */

struct Model {
    int E = 0, P = 1, n = 0;
    bool ancestral_aware = false, dephase = false;
    double L = 0, mu = 0, rho = 0;
    std::vector<double> T;       /* epoch start times */
    std::vector<double> inv2N;   /* 1/(2 N_e) */
    std::vector<double> Hc;      /* cumulative coalescence intensity at the epoch starts (one population) */
    std::vector<int> recflags;
    std::vector<double> lags;
    /* structured models (P > 1): per-epoch, per-population tables (scrm Model::population_size,
     * migration_rate, single_mig_pop) */
    std::vector<double> inv2Np;      /* [E*P]   1/(2 N_e,p) */
    std::vector<double> Mrate;       /* [E*P*P] backward migration rate p -> q per generation */
    std::vector<double> Mtot;        /* [E*P]   sum_q Mrate[e][p][q], summed q ascending */
    std::vector<int> jmap;           /* [E*P]   population after the fixed-time moves at the start of epoch e (-ej) */
    std::vector<double> Cc, Cm;      /* [E*P]   integrals of inv2Np / Mtot from 0 to the epoch starts (the one-population Hc, per population) */
    std::vector<double> Tjoin;       /* [E]     start of the next epoch after e whose jmap moves a population, +inf if none */
    std::vector<int> sample_pop;     /* [n] */
    /* variational-Bayes weight factors exp_digamma(c)/c per event (particle.cpp:266-272); empty = off */
    std::vector<double> vb_coal;     /* [E*P] */
    std::vector<double> vb_mig;      /* [E*P*P] */
    /* focused sampling (Model::bias_heights / bias_strengths of the scrm fork; particle.cpp:1020-1050) */
    bool biased = false;
    std::vector<double> bias_H;      /* 0, h1..hk, +inf */
    std::vector<double> bias_S;      /* k+1 strengths */
    std::vector<double> app_delays;  /* Model::application_delays (smcsmc.cpp:306-307) */
    int delay_type = 0;              /* PfParam::ResampleDelayType: 0 recombination height, 1 coalescence, 2 coal/migr */
    /* recombination guide (pfparam.hpp:96-223 RecombinationBias): piecewise-constant sampling rate along the sequence and
     * relative rates per leaf; rho stays the true rate */
    bool guided = false;
    std::vector<double> seg_pos;     /* [K] segment starts, seg_pos[0] == 0 */
    std::vector<double> seg_rho;     /* [K] */
    std::vector<double> leaf_rate;   /* [K*n] */
    int epoch_of(double t) const {
        int e = 0;
        while (e + 1 < E && T[e + 1] <= t) ++e;
        return e;
    }
    double epoch_end(int e) const { return e + 1 < E ? T[e + 1] : HUGE_VAL; }
};

/* ------------------------------------------------------------------ event records
 * reference: coalevent.hpp:71-369 (EvolutionaryEvent), arena.cpp (allocator; here a free list) */

struct Ev {
    double t0, t1;      /* start_height, end_height */
    double x0, x1;      /* recomb: [start_base,end_base]; coal: x0 = x1 = position */
    Ev* parent;
    double acc;         /* posterior_ */
    int32_t arrived;    /* children_updated_ */
    int32_t refs;       /* ref_counter_ */
    int16_t weight;     /* number of contemporaries */
    int8_t kind;        /* 0 recombination opportunity, 1 coalescence opportunity */
    int8_t event;       /* 1 if an event sits on this record (recomb: at (x0, ev_t); coal: at t1); 2 migration at t1 */
    int8_t dead;
    int8_t pop;         /* coal/migr records: population of the active lineage */
    int8_t mig_to;      /* event == 2: destination population of the migration event */
    double ev_t;        /* recombination event height (coalevent.hpp:170-176) */
    uint64_t desc;      /* recombination event: samples below the cut branch (descendants.hpp:22-33), bit i = sample i */
};

struct Pool {
    std::vector<Ev*> blocks;
    Ev* free_list = nullptr;
    int64_t n_alloc = 0, n_live = 0;
    Ev* get() {
        if (!free_list) {
            Ev* b = (Ev*)std::malloc(sizeof(Ev) * 65536);
            blocks.push_back(b);
            for (int i = 0; i < 65536; ++i) { b[i].parent = free_list; free_list = &b[i]; }
        }
        Ev* e = free_list;
        free_list = e->parent;
        ++n_alloc; ++n_live;
        return e;
    }
    void put(Ev* e) { e->parent = free_list; free_list = e; --n_live; }
    ~Pool() { for (Ev* b : blocks) std::free(b); }
};

/* ------------------------------------------------------------------ particle */

struct Tree {
    /* local tree over n leaves, internal nodes kept sorted by height:
     * S[r] = height of the internal node of rank r; C[r][0..1] = its children,
     * child id < n : leaf (sample index), id >= n : internal node of rank id-n. */
    double S[NMAX - 1];
    int8_t C[NMAX - 1][2];
    /* structured models: population in which each coalescence happened, and the migration events on the
     * branches of the local tree, sorted by time.  Event m sits on the branch above node id Mb[m] (the
     * child end of the branch) and moves the lineage to population Mq[m] at time Mt[m].  scrm keeps
     * these as unary "migrating" nodes (Node::is_migrating); a list keeps the rank-sorted binary tree
     * -- and with it likelihood / branch-length code -- untouched. */
    int8_t Pn[NMAX - 1];
    int nm = 0;
    double Mt[MMAX];
    int8_t Mb[MMAX], Mq[MMAX];
};

/* a tree-modifying event of the pseudo-epoch of particle.cpp:240-249, 292-298, 379-389 (RECORD_TREE_EVENT): the
 * list is immutable and shared between a particle and its copies, as the reference's ref-counted event chains are */
struct TreeEv {
    int kind;                 /* 0 recombination (R), 1 coalescence (C), 2 migration (M) */
    double x, t;
    uint32_t desc;
    int from_pop, to_pop;     /* population of a C / M event, destination of an M event; -1 otherwise (pc.cpp:541-548) */
    std::shared_ptr<const TreeEv> parent;
};

struct Particle {
    std::shared_ptr<const TreeEv> tree_head;   /* eventTrees[tree_epoch] (particle.cpp:242, 294, 381) */
    Tree tr;
    double w_post, w_pilot;
    double next_base;
    double x_mark;          /* start of the current recombination-opportunity stretch */
    int mark_limit;         /* max_epoch_to_record_ in force when the stretch was opened */
    double Ltree;
    double lookahead = 1.0;                  /* lookahead_weight_ (particle.hpp:239) */
    int ridx = 0;                            /* _current_seq_idx: guide segment the particle is in (particle.hpp:177-178) */
    double total_delayed = 1.0;              /* total_delayed_adjustment_ */
    int dcount = 0;                          /* pending DelayedFactors (particle.hpp:248: a std::priority_queue) */
    std::vector<double> dpos, dfac, ddelta;
    std::vector<int> dk;
    std::vector<Ev*> head;  /* eventTrees[epoch] (particle.hpp:235) */
    std::vector<Ev*> open;  /* the open rectangles of the current stretch (heads of their chains) */
};

struct SlotRng { uint64_t ctr; double ebuf; };

struct Filter {
    Model M;
    int64_t Np;
    double ess_fraction;
    uint64_t seed;
    std::vector<Particle> parts;
    std::vector<SlotRng> rng;
    Pool pool;
    double logl = 0;
    double cur_pos = 0;           /* site_where_weight_was_updated_ (identical for all particles) */
    int cur_limit = 0;
    int64_t n_resample = 0;
    int64_t n_recomb = 0;
    /* count model (count.hpp) */
    std::vector<double> coal_count, coal_opp, coal_w2;   /* [E*P] */
    std::vector<double> rec_count, rec_opp, rec_w2, counted_to;
    std::vector<double> mig_count;                       /* [E*P*P] */
    std::vector<double> mig_opp, mig_w2;                 /* [E*P] */
    /* local recombination map (count.hpp:101-102, 115): differential opportunity and per-sample / time / log-time counts
     * per 100-bp interval */
    bool local_map = false;
    std::vector<double> local_opp;
    std::vector<std::vector<double>> local_cnt;
    double delayed_opp = 0, delayed_count = 0;
    /* trace */
    std::vector<double> tr_T, tr_ess, tr_logl;
    std::vector<int32_t> tr_flag;
    int32_t max_trace_events;
    std::vector<int32_t> ev_seg;
    std::vector<std::vector<int32_t>> ev_parents;
    int64_t seg_done = 0;

    /* auxiliary particle filter */
    int apf = 0, la_D = 0, la_Q = 0;
    std::vector<double> la_fsd, la_rmr, la_ddist, la_split, la_q, la_tbl;
    std::vector<int8_t> la_unph, la_didx, la_salleles;
    std::vector<int32_t> la_nd, la_sk;
    double la_mean_tbl = 0;
    uint32_t stream = 0;          /* Philox stream id: 0 particle filter, 2 lag calibration, 3 branch-length quantiles */
    bool record_events = true;
    double last_sp = 0; bool last_changed = false;
    uint64_t last_desc = 0;       /* samples below the branch cut by the last genealogy update */
    /* product of the variational-Bayes factors of the events of one walk; applied to the weights when the walk is
     * over (the reference multiplies event by event: same product, rounding aside) */
    double upd_fac = 1.0;
    void apply_vb(Particle& p) { if (!M.vb_coal.empty()) { p.w_post *= upd_fac; p.w_pilot *= upd_fac; } upd_fac = 1.0; }
    double last_iw = 1.0, last_tc = 0.0, last_first_event = 0.0;
    bool record_trees = false;      /* -arg (pfparam.cpp:353-357) */
    int mig_cap = 96;               /* migration events a local tree may hold (capacity of the device path) */
    int delay_cap = DCAP_DEFAULT;   /* pending delayed factors a particle may hold (capacity of the device path) */
    bool delay_evict = false;
    int64_t n_delay_evict = 0;      /* factors applied ahead of their position to make room (delay_evict only) */
    int delay_peak = 0;             /* most factors any particle ever had pending */
    void push_tree_event(Particle& p, int kind, double x, double t, uint32_t desc, int from_pop = -1, int to_pop = -1) {
        auto ev = std::make_shared<TreeEv>();
        ev->kind = kind; ev->x = x; ev->t = t; ev->desc = desc; ev->parent = p.tree_head;
        ev->from_pop = kind == 1 && from_pop < 0 ? 0 : from_pop;     /* one population: every coalescence is in population 0 */
        ev->to_pop = to_pop;
        p.tree_head = ev;
    }
    /* -arg with several populations: the samples whose lineage the two active lineages of the walk in progress are (what
     * get_descendants(active_node(i)) returns in particle.cpp:292-298) */
    uint32_t tree_fl_desc = 0, tree_rt_desc = 0;
    /* samples below node id of tree t */
    uint32_t desc_mask(const Tree& t, int id) const {
        if (id < M.n) return 1u << id;
        return desc_mask(t, t.C[id - M.n][0]) | desc_mask(t, t.C[id - M.n][1]);
    }
    double last_rbiw = 1.0;       /* recombination_bias_importance_weight_ (particle.cpp:1113-1121) */
    int64_t slot_override = -1;   /* calibration: RNG state lives in rng[0], stream keyed by the replicate index */
    /* The four uniforms of one genealogy update of the one-population engine (cut point, waiting-time refresh,
     * re-attachment slot, next recombination position) are the two halves of two consecutive Philox blocks, drawn
     * together when the update starts; the counter advances by two per update whether or not the fourth is used.
     * Every other draw takes the first half of its own block. */
    double uq[4];
    int uq_n = 0, uq_i = 0;
    void prefetch_update_uniforms(int64_t slot) {
        const uint32_t sl = (uint32_t)(slot_override >= 0 ? slot_override : slot);
        philox_pair(seed, sl, stream, rng[slot].ctr, &uq[0], &uq[1]);
        philox_pair(seed, sl, stream, rng[slot].ctr + 1, &uq[2], &uq[3]);
        rng[slot].ctr += 2;
        uq_n = 4; uq_i = 0;
    }
    void drop_update_uniforms() { uq_n = uq_i = 0; }
    double uni(int64_t slot) {
        if (uq_i < uq_n) return uq[uq_i++];
        return philox_uniform(seed, (uint32_t)(slot_override >= 0 ? slot_override : slot), stream, rng[slot].ctr++);
    }

    /* ---------------- tree helpers ---------------- */
    inline double node_h(const Tree& t, int id) const { return id < M.n ? 0.0 : t.S[id - M.n]; }

    /* local tree length by time slices: sum_r (n-r) * (S[r]-S[r-1])   (Forest::getLocalTreeLength) */
    double tree_length(const Tree& t, int nleaves) const {
        double acc = 0.0, prev = 0.0;
        for (int r = 0; r < nleaves - 1; ++r) {
            acc += (double)(nleaves - r) * (t.S[r] - prev);
            prev = t.S[r];
        }
        return acc;
    }

    /* enumerate the lineages of the tree (ni internal nodes stored) that cross time t, in
     * canonical order: parent rank ascending, child 0 then child 1.  Returns the count and
     * writes the idx-th slot to (pr, ps).  Lineages are "slots": child pointers whose child
     * lies at or below t while the parent lies above t. */
    int lineages_at(const Tree& t, int ni, double time, int want, int* pr, int* ps) const {
        int R = 0;
        while (R < ni && t.S[R] <= time) ++R;
        int cnt = 0;
        for (int r = R; r < ni; ++r)
            for (int s = 0; s < 2; ++s) {
                int id = t.C[r][s];
                if (id < M.n || id - M.n < R) {
                    if (cnt == want) { *pr = r; *ps = s; }
                    ++cnt;
                }
            }
        return cnt;
    }

    /* remove internal node of rank rp from a tree with ni internal nodes; its parent slot
     * (if any) is re-pointed to `sib`.  Tracked ids *a,*b are relabelled alongside. */
    void remove_rank(Tree& t, int ni, int rp, int sib, int* a, int* b) const {
        const int n = M.n;
        const int pid = n + rp;
        for (int r = rp + 1; r < ni; ++r)
            for (int s = 0; s < 2; ++s)
                if (t.C[r][s] == pid) t.C[r][s] = (int8_t)sib;
        for (int r = rp; r + 1 < ni; ++r) {
            t.S[r] = t.S[r + 1];
            t.C[r][0] = t.C[r + 1][0];
            t.C[r][1] = t.C[r + 1][1];
        }
        for (int r = 0; r < ni - 1; ++r)
            for (int s = 0; s < 2; ++s)
                if (t.C[r][s] > pid) t.C[r][s] -= 1;
        if (*a > pid) *a -= 1;
        if (*b > pid) *b -= 1;
    }

    /* insert a new internal node at height h joining floating id `fl` and the lineage in slot
     * (pr,ps) -- or the tree root `root_id` if pr < 0 -- into a tree with ni internal nodes. */
    void insert_node(Tree& t, int ni, double h, int fl, int pr, int ps, int root_id) const {
        const int n = M.n;
        int rn = 0;
        while (rn < ni && t.S[rn] <= h) ++rn;
        const int nid = n + rn;
        for (int r = 0; r < ni; ++r)
            for (int s = 0; s < 2; ++s)
                if (t.C[r][s] >= nid) t.C[r][s] += 1;
        if (fl >= nid) fl += 1;
        if (root_id >= nid) root_id += 1;
        for (int r = ni; r > rn; --r) {
            t.S[r] = t.S[r - 1];
            t.C[r][0] = t.C[r - 1][0];
            t.C[r][1] = t.C[r - 1][1];
        }
        int target;
        if (pr >= 0) {
            if (pr >= rn) pr += 1;
            target = t.C[pr][ps];
            t.C[pr][ps] = (int8_t)nid;
        } else {
            target = root_id;
        }
        t.S[rn] = h;
        t.C[rn][0] = (int8_t)fl;
        t.C[rn][1] = (int8_t)target;
    }

    /* ---------------- event recording (particle.cpp:193-390) ---------------- */
    Ev* new_event(Particle& p, int epoch, int kind, double t0, double t1, double x0, double x1, int weight) {
        Ev* e = pool.get();
        e->t0 = t0; e->t1 = t1; e->x0 = x0; e->x1 = x1;
        e->acc = 0; e->arrived = 0; e->refs = 1;
        e->weight = (int16_t)weight; e->kind = (int8_t)kind; e->event = 0; e->dead = 0; e->ev_t = -1; e->pop = 0; e->mig_to = 0; e->desc = 0;
        e->parent = p.head[epoch];      /* add_leaf_to_tree: coalevent.hpp:288-303 (takes over the head's reference) */
        p.head[epoch] = e;
        return e;
    }
    void release(Ev* e) {
        while (e && --e->refs == 0) {
            Ev* par = e->parent;
            pool.put(e);
            e = par;
        }
    }

    /* ================= structured models (P > 1) =================
     * Reconstructed from the control flow of particle.cpp:1266-1521 (sampleNextGenealogyWithoutImplementing and
     * its dontImplement* helpers, which mirror the scrm fork's implementing versions) with the rates of
     * an active lineage in population p coalesces at rate #contemporaries(p)/(2 N_p), migrates to q
     * at rate M[p][q]; two active lineages in the same population coalesce pairwise at rate 1/(2 N_p);
     * fixed-time moves (-ej) are applied when an active lineage crosses the epoch boundary
     * (dontImplementFixedTimeEvent, particle.cpp:1387-1418; deterministic moves only). */

    int pop_base(const Tree& t, int id) const { return id < M.n ? M.sample_pop[id] : t.Pn[id - M.n]; }
    /* population of the lineage above node `id` at time `time` */
    int pop_at(const Tree& t, int id, double time) const {
        int pop = pop_base(t, id);
        for (int m = 0; m < t.nm; ++m)
            if (t.Mb[m] == id && t.Mt[m] <= time) pop = t.Mq[m];
        return pop;
    }
    /* lineages_at restricted to the lineages that are in population `pop` at `time` */
    int lineages_in_pop(const Tree& t, int ni, double time, int pop, int want, int* pr, int* ps) const {
        int R = 0;
        while (R < ni && t.S[R] <= time) ++R;
        int cnt = 0;
        for (int r = R; r < ni; ++r)
            for (int s = 0; s < 2; ++s) {
                int id = t.C[r][s];
                if ((id < M.n || id - M.n < R) && pop_at(t, id, time) == pop) {
                    if (cnt == want) { *pr = r; *ps = s; }
                    ++cnt;
                }
            }
        return cnt;
    }
    void ev_insert(Tree& t, double time, int branch, int newpop) const {
        if (t.nm >= mig_cap) throw std::runtime_error("too many migration events on one local tree");
        int m = t.nm;
        while (m > 0 && t.Mt[m - 1] > time) {
            t.Mt[m] = t.Mt[m - 1]; t.Mb[m] = t.Mb[m - 1]; t.Mq[m] = t.Mq[m - 1];
            --m;
        }
        t.Mt[m] = time; t.Mb[m] = (int8_t)branch; t.Mq[m] = (int8_t)newpop;
        ++t.nm;
    }
    /* drop the events on branch `id` later than `tmin` */
    void ev_drop_above(Tree& t, int id, double tmin) const {
        int o = 0;
        for (int m = 0; m < t.nm; ++m) {
            if (t.Mb[m] == id && t.Mt[m] > tmin) continue;
            t.Mt[o] = t.Mt[m]; t.Mb[o] = t.Mb[m]; t.Mq[o] = t.Mq[m];
            ++o;
        }
        t.nm = o;
    }
    /* remove_rank that also maintains node populations and the migration list: the events on the
     * removed node's own branch pass to the sibling lineage that takes its place */
    void mp_remove_rank(Tree& t, int ni, int rp, int sib, int* a, int* b) const {
        const int pid = M.n + rp;
        for (int m = 0; m < t.nm; ++m) {
            if (t.Mb[m] == pid) t.Mb[m] = (int8_t)sib;
            if (t.Mb[m] > pid) t.Mb[m] -= 1;
        }
        for (int r = rp; r + 1 < ni; ++r) t.Pn[r] = t.Pn[r + 1];
        remove_rank(t, ni, rp, sib, a, b);
    }
    /* insert_node that also maintains node populations and the migration list: the part of the target
     * branch above the new node becomes the new node's branch (or vanishes above a new root) */
    void mp_insert_node(Tree& t, int ni, double h, int* fl, int pr, int ps, int root_id, int node_pop) const {
        const int n = M.n;
        int rn = 0;
        while (rn < ni && t.S[rn] <= h) ++rn;
        const int nid = n + rn;
        int target = pr >= 0 ? t.C[pr][ps] : root_id;
        for (int m = 0; m < t.nm; ++m)
            if (t.Mb[m] >= nid) t.Mb[m] += 1;
        if (target >= nid) target += 1;
        if (pr >= 0) {
            for (int m = 0; m < t.nm; ++m)
                if (t.Mb[m] == target && t.Mt[m] > h) t.Mb[m] = (int8_t)nid;
        } else {
            ev_drop_above(t, target, h);
        }
        for (int r = ni; r > rn; --r) t.Pn[r] = t.Pn[r - 1];
        t.Pn[rn] = (int8_t)node_pop;
        insert_node(t, ni, h, *fl, pr, ps, root_id);
        if (*fl >= nid) *fl += 1;
    }

    struct Walk {
        double tc;
        int pf, pr;                       /* populations of the floating / root lineage at tc */
        int npath, nrpath;
        double pt[MMAX], rt[MMAX];        /* migration events picked up on the way: floating / root lineage */
        int8_t pq[MMAX], rq[MMAX];
        int weight;                       /* coalescence partners at tc (consistency check) */
        double tfirst;                    /* first sampled event of the walk, migration or coalescence (first_event_height_) */
    };

    /* The floating lineage starts at height h in population pf0 and moves up through the tree `t`
     * (ni internal nodes; root_id its top node, or the single leaf).  Above the root the root's own
     * lineage is the second active lineage.  One unit exponential is consumed across the intervals
     * (sampleExpoLimit), the kind of event is drawn with one uniform.
     *
     * D14 (arithmetic, not distribution).  The reference stops at every epoch boundary, node and event and
     * subtracts (length x rate) from the budget (particle.cpp:230-300 with the TimeIntervalIterator of scrm).
     * Between two changes of the configuration (next node, next migration event on the tree, next epoch with a
     * fixed-time move) the lineages' populations and the number of partners are constant, so the hazard over
     * that stretch is a difference of the cumulative intensities Cc / Cm tabulated at the epoch starts -- the
     * same step D12 takes for one population with Hc.  The budget is compared with that difference once per
     * stretch; in the stretch in which it runs out the epoch of the event is found by comparing the budget with the
     * same difference taken to the epoch starts of the stretch.  The event records are still cut per epoch (one Ev
     * per epoch of the stretch), as record_all_event makes them. */
    void mp_coalesce(int64_t slot, Particle* rec_p, const Tree& t, int ni, int root_id, double h, int pf0,
                     double x, int limit, Walk& W) {
        SlotRng& g = rng[slot];
        const int P = M.P;
        const double Hr = node_h(t, root_id);
        double tt = h;
        int e = M.epoch_of(tt);
        int i = 0, j = 0;
        while (i < ni && t.S[i] <= tt) ++i;
        while (j < t.nm && t.Mt[j] <= tt) ++j;
        int pf = pf0, pr = pop_base(t, root_id);
        W.npath = W.nrpath = 0;
        W.tfirst = -1.0;
        auto ci = [&](int q, int ee, double tm) { return M.Cc[ee * P + q] + (tm - M.T[ee]) * M.inv2Np[ee * P + q]; };
        auto cm = [&](int q, int ee, double tm) { return M.Cm[ee * P + q] + (tm - M.T[ee]) * M.Mtot[ee * P + q]; };
        /* record_all_event for [a, b) of the walk, cut at the epoch boundaries; the event (if any) sits at b */
        auto record = [&](bool root_active, int weight, double a, double b, int ea, int kind, int to) {
            if (!(rec_p && record_events)) return;
            for (int ee = ea; ee < M.E; ++ee) {
                double lo = std::max(a, M.T[ee]), hi = std::min(b, M.epoch_end(ee));
                const bool last = !(M.epoch_end(ee) <= b) || ee + 1 == M.E;
                if ((M.recflags[ee] & REC_COALMIGR) && ee <= limit && (hi > lo || (last && kind != 0))) {
                    if (hi < lo) hi = lo;
                    Ev* ev = new_event(*rec_p, ee, 1, lo, hi, x, x, weight);
                    ev->pop = (int8_t)pf;
                    if (last && kind == 1) ev->event = 1;
                    if (last && kind == 2) { ev->event = 2; ev->mig_to = (int8_t)to; }
                    if (root_active) {
                        /* second active node: no contemporaries above the root, migration opportunity only */
                        Ev* ev2 = new_event(*rec_p, ee, 1, lo, hi, x, x, 0);
                        ev2->pop = (int8_t)pr;
                        if (last && kind == 3) { ev2->event = 2; ev2->mig_to = (int8_t)to; }
                    }
                }
                if (last) break;
            }
        };
        for (;;) {
            const bool root_active = tt >= Hr;
            double tn_node = i < ni ? t.S[i] : HUGE_VAL;
            double tn_mig = j < t.nm ? t.Mt[j] : HUGE_VAL;
            const double tj = M.Tjoin[e];
            double tn = std::min(std::min(tn_node, tn_mig), tj);
            int dummy_r = 0, dummy_s = 0;
            int k = lineages_in_pop(t, ni, tt, pf, -1, &dummy_r, &dummy_s);
            int weight = k + ((root_active && pr == pf) ? 1 : 0);      /* particle.cpp:255-260, 277 */
            int en = e;
            bool quiet = false;
            if (tn < HUGE_VAL) {
                while (en + 1 < M.E && M.T[en + 1] <= tn) ++en;
                double need = (double)weight * (ci(pf, en, tn) - ci(pf, e, tt)) + (cm(pf, en, tn) - cm(pf, e, tt));
                if (root_active) need = need + (cm(pr, en, tn) - cm(pr, e, tt));
                if (g.ebuf > need) { g.ebuf -= need; quiet = true; }
            }
            if (!quiet) {
                /* the epoch of the event: the hazard from tt to the start of epoch k is the same difference of cumulative
                 * intensities as `need` (it ascends with k), so the event falls into the last epoch of the stretch whose
                 * start the budget still reaches; one division places it there */
                const double f0c = ci(pf, e, tt), f0m = cm(pf, e, tt), f0r = root_active ? cm(pr, e, tt) : 0.0;
                const int elim = tn < HUGE_VAL ? en : M.E - 1;
                int ee = e;
                double gee = 0.0;
                while (ee < elim) {
                    double gk = (double)weight * (M.Cc[(ee + 1) * P + pf] - f0c) + (M.Cm[(ee + 1) * P + pf] - f0m);
                    if (root_active) gk = gk + (M.Cm[(ee + 1) * P + pr] - f0r);
                    if (!(g.ebuf > gk)) break;
                    gee = gk;
                    ++ee;
                }
                const double rc = (double)weight * M.inv2Np[ee * P + pf];
                const double rmf = M.Mtot[ee * P + pf];
                const double rmr = root_active ? M.Mtot[ee * P + pr] : 0.0;
                const double lam = (rc + rmf) + rmr;
                if (lam == 0.0) throw std::logic_error("No final coalescence event was sampled!");
                double t1 = ee == e ? tt + g.ebuf / lam : M.T[ee] + (g.ebuf - gee) / lam;
                {
                    const double up = std::min(M.epoch_end(ee), tn);
                    if (t1 > up) t1 = up;
                }
                {
                    int kind = 0, to = 0;          /* 1 coalescence, 2 floating lineage migrates, 3 root lineage migrates */
                    double v = uni(slot) * lam;
                    if (v < rc || (rmf == 0.0 && rmr == 0.0)) kind = 1;
                    else {
                        v -= rc;
                        int from;
                        if (v < rmf || rmr == 0.0) { kind = 2; from = pf; }
                        else { kind = 3; from = pr; v -= rmf; }
                        to = -1;
                        for (int q = 0; q < P; ++q) {
                            double m = M.Mrate[(ee * P + from) * P + q];
                            if (q == from || m == 0.0) continue;
                            to = q;
                            if (v < m) break;
                            v -= m;
                        }
                    }
                    record(root_active, weight, tt, t1, e, kind, to);
                    if (record_trees && rec_p && kind != 1)     /* the M line of printTrees (particle.cpp:292-298) */
                        push_tree_event(*rec_p, 2, x, t1, kind == 2 ? tree_fl_desc : tree_rt_desc, kind == 2 ? pf : pr, to);
                    if (W.tfirst < 0.0) W.tfirst = t1;          /* particle.cpp:263-264 */
                    g.ebuf = -smc_log(uni(slot));
                    if (rec_p && !M.vb_coal.empty())            /* adjustWeights(exp_digamma(c)/c), particle.cpp:266-272 */
                        upd_fac *= kind == 1 ? M.vb_coal[ee * P + pf] : M.vb_mig[(ee * P + (kind == 2 ? pf : pr)) * P + to];
                    if (kind == 1) {
                        W.tc = t1; W.pf = pf; W.pr = pr; W.weight = weight;
                        return;
                    }
                    if (kind == 2) {
                        if (W.npath >= MMAX) throw std::runtime_error("too many migration events on one local tree");
                        W.pt[W.npath] = t1; W.pq[W.npath] = (int8_t)to; ++W.npath;
                        pf = to;
                    } else {
                        if (W.nrpath >= MMAX) throw std::runtime_error("too many migration events on one local tree");
                        W.rt[W.nrpath] = t1; W.rq[W.nrpath] = (int8_t)to; ++W.nrpath;
                        pr = to;
                    }
                    tt = t1;
                    e = ee;
                    continue;
                }
            }
            record(root_active, weight, tt, tn, e, 0, 0);
            const bool at_join = !(tn < tj);
            tt = tn;
            e = en;
            while (i < ni && t.S[i] <= tt) ++i;
            while (j < t.nm && t.Mt[j] <= tt) ++j;
            if (at_join) {
                /* fixed-time moves at the start of epoch e */
                int q = M.jmap[e * P + pf];
                if (q != pf) {
                    if (W.npath >= MMAX) throw std::runtime_error("too many migration events on one local tree");
                    W.pt[W.npath] = tt; W.pq[W.npath] = (int8_t)q; ++W.npath;
                    pf = q;
                }
                if (tt >= Hr) {
                    int qr = M.jmap[e * P + pr];
                    if (qr != pr) {
                        if (W.nrpath >= MMAX) throw std::runtime_error("too many migration events on one local tree");
                        W.rt[W.nrpath] = tt; W.rq[W.nrpath] = (int8_t)qr; ++W.nrpath;
                        pr = qr;
                    }
                }
            }
        }
    }

    void mp_build_initial_tree(int64_t slot, Particle& p) {
        const int n = M.n;
        Tree& t = p.tr;
        t.nm = 0;
        int root = 0;
        for (int i = 1; i < n; ++i) {
            int ni = i - 1;
            Walk W;
            tree_fl_desc = 1u << i; tree_rt_desc = (1u << i) - 1u;
            mp_coalesce(slot, &p, t, ni, root, 0.0, M.sample_pop[i], 0.0, M.E - 1, W);
            apply_vb(p);
            double tc = W.tc;
            for (int m = 0; m < W.nrpath; ++m) ev_insert(t, W.rt[m], root, W.rq[m]);
            int pr = -1, ps = 0;
            int nslots = lineages_in_pop(t, ni, tc, W.pf, -1, &pr, &ps);
            bool has_root = tc >= node_h(t, root) && pop_at(t, root, tc) == W.pf;
            int k = nslots + (has_root ? 1 : 0);
            if (k != W.weight) throw std::logic_error("initial tree: coalescence partners inconsistent");
            double u = uni(slot);
            int idx = std::min((int)(u * (double)k), k - 1);
            int fl = i;
            if (idx < nslots) {
                lineages_in_pop(t, ni, tc, W.pf, idx, &pr, &ps);
                if (record_trees) push_tree_event(p, 1, 0.0, tc, (1u << i) | desc_mask(t, t.C[pr][ps]), W.pf);
                mp_insert_node(t, ni, tc, &fl, pr, ps, root, W.pf);
            } else {
                if (record_trees) push_tree_event(p, 1, 0.0, tc, (2u << i) - 1u, W.pf);
                mp_insert_node(t, ni, tc, &fl, -1, 0, root, W.pf);
            }
            for (int m = 0; m < W.npath; ++m) ev_insert(t, W.pt[m], fl, W.pq[m]);
            root = n + ni;
        }
        p.Ltree = tree_length(t, n);
    }

    /* the part of genealogy_update after the recombination point (slot (rp,sb), height h) is known */
    void mp_genealogy_rest(int64_t slot, Particle& p, double x, int limit, int rp, int sb, double h) {
        const int n = M.n;
        Tree& t = p.tr;
        int b_id = t.C[rp][sb], s_id = t.C[rp][1 - sb];
        const int pf0 = pop_at(t, b_id, h);
        Walk W;
        const uint32_t cut = record_trees ? desc_mask(t, b_id) : 0u;
        tree_fl_desc = cut; tree_rt_desc = ((1u << n) - 1u) & ~cut;
        mp_coalesce(slot, &p, t, n - 1, n + n - 2, h, pf0, x, limit, W);
        const double tc = W.tc;
        last_tc = tc;
        last_first_event = W.tfirst;
        const double Sp = t.S[rp];
        const int p_pop = t.Pn[rp];
        const bool p_was_root = (rp == n - 2);
        /* the stub: what remains of the cut branch above the cut, with its migration events */
        int nstub = 0; double st_t[MMAX]; int8_t st_q[MMAX];
        for (int m = 0; m < t.nm; ++m)
            if (t.Mb[m] == b_id && t.Mt[m] > h) { st_t[nstub] = t.Mt[m]; st_q[nstub] = t.Mq[m]; ++nstub; }
        ev_drop_above(t, b_id, h);
        mp_remove_rank(t, n - 1, rp, s_id, &b_id, &s_id);
        int ni = n - 2;
        int troot = p_was_root ? s_id : n + (ni - 1);
        for (int m = 0; m < W.nrpath; ++m) ev_insert(t, W.rt[m], troot, W.rq[m]);
        int pr = -1, ps = 0;
        int nslots = lineages_in_pop(t, ni, tc, W.pf, -1, &pr, &ps);
        bool has_root = tc >= node_h(t, troot) && pop_at(t, troot, tc) == W.pf;
        int stub_pop = pf0;
        for (int m = 0; m < nstub; ++m) if (st_t[m] <= tc) stub_pop = st_q[m];
        bool has_stub = tc < Sp && stub_pop == W.pf;
        int k = nslots + (has_root ? 1 : 0) + (has_stub ? 1 : 0);
        if (k != W.weight) throw std::logic_error("genealogy update: coalescence partners inconsistent");
        double u = uni(slot);
        int idx = std::min((int)(u * (double)k), k - 1);
        last_sp = Sp;
        last_changed = !(has_stub && idx == k - 1);
        if (record_trees) {
            /* as in the one-population update: C with the samples below the node the lineage creates, then R */
            uint32_t dn = cut;
            if (idx < nslots) {
                int qr = -1, qs = 0;
                lineages_in_pop(t, ni, tc, W.pf, idx, &qr, &qs);
                dn = cut | desc_mask(t, t.C[qr][qs]);
            } else if (has_root && idx == nslots) {
                dn = (1u << n) - 1u;
            }
            push_tree_event(p, 1, x, tc, dn, W.pf);
            push_tree_event(p, 0, x, h, cut);
        }
        if (idx < nslots) {
            lineages_in_pop(t, ni, tc, W.pf, idx, &pr, &ps);
            mp_insert_node(t, ni, tc, &b_id, pr, ps, troot, W.pf);
        } else if (has_root && idx == nslots) {
            mp_insert_node(t, ni, tc, &b_id, -1, 0, troot, W.pf);
        } else {
            /* back into its own stub: the tree keeps its shape, the cut branch swaps the events between the
             * cut and tc for the ones picked up on the way */
            if (p_was_root) {
                mp_insert_node(t, ni, Sp, &b_id, -1, 0, troot, p_pop);
            } else {
                int want = -1, c = 0, R = 0;
                while (R < ni && t.S[R] <= Sp) ++R;
                int fr = -1, fs = 0;
                for (int rr = R; rr < ni && want < 0; ++rr)
                    for (int s = 0; s < 2 && want < 0; ++s) {
                        int id = t.C[rr][s];
                        if (id < n || id - n < R) {
                            if (id == s_id) { want = c; fr = rr; fs = s; }
                            ++c;
                        }
                    }
                mp_insert_node(t, ni, Sp, &b_id, fr, fs, troot, p_pop);
            }
            for (int m = 0; m < nstub; ++m) if (st_t[m] > tc) ev_insert(t, st_t[m], b_id, st_q[m]);
        }
        for (int m = 0; m < W.npath; ++m) ev_insert(t, W.pt[m], b_id, W.pq[m]);
        ev_drop_above(t, n + n - 2, -1.0);      /* nothing is kept above the root of the local tree */
        p.Ltree = tree_length(t, n);
        ++n_recomb;
    }

    /* record_recomb_extension (particle.cpp:305-357): one rectangle per (tree time-slice x epoch)
     * with weight = number of local contemporaries; opened at x = p.x_mark, closed later (D6). */
    void open_stretch(Particle& p, double x, int limit) {
        p.x_mark = x;
        p.mark_limit = limit;
        p.open.clear();
        const int n = M.n;
        double prev = 0.0;
        for (int r = 0; r < n - 1; ++r) {
            double top = p.tr.S[r];
            int k = n - r;
            double t = prev;
            int e = M.epoch_of(t);
            while (t < top) {
                double t1 = std::min(top, M.epoch_end(e));
                if ((M.recflags[e] & REC_RECOMB) && e <= limit)
                    p.open.push_back(new_event(p, e, 0, t, t1, x, HUGE_VAL, k));
                t = t1;
                if (t < top) ++e;
            }
            prev = top;
        }
    }
    void close_stretch(Particle& p, double x) {
        for (Ev* e : p.open) e->x1 = x;
        p.open.clear();
    }

    /* ---------------- the floating-lineage coalescence (SMC') ----------------
     * Reconstructed Forest::sampleCoalescences for one population; interval walk mirrors
     * particle.cpp:1325-1382, rates per SURVEY 8c, exponential buffer per SURVEY A12.
     * Sh[0..ns) are the sorted heights that delimit the lineage count k(t) = nl - #{Sh <= t}
     * (1 above the top).  Records one coal-opportunity record per interval (record_all_event,
     * particle.cpp:251-300) at position x, and returns the coalescence time. */
    double coalesce_up(int64_t slot, Particle* rec_p, const double* Sh, int ns, int nl, double h, double x,
                       int limit) {
        SlotRng& g = rng[slot];
        /* The time of the event.  The exponential waiting time of the interval walk (rate k(t)/(2N(t)), one unit
         * exponential carried across intervals) is evaluated on the cumulative intensity
         * Hc(t) = int_0^t ds/(2N(s)), tabulated at the epoch starts: between two nodes k is constant, so the budget is
         * compared with k (Hc(next node) - Hc(t)) once per node, and the event sits where Hc reaches
         * Hc(t) + budget / k.  Same distribution as drawing interval by interval; the device does exactly this
         * arithmetic (coalesce_up in pf_device.h, r_coalesce_up in pf_tree_reg.h). */
        const int e0 = M.epoch_of(h);
        int i0 = 0;
        while (i0 < ns && Sh[i0] <= h) ++i0;
        double Hc = M.Hc[e0] + (h - M.T[e0]) * M.inv2N[e0];
        int i = i0;
        double lower = h, kd;
        for (;;) {
            if (i >= ns) { kd = 1.0; break; }
            kd = (double)(nl - i);
            const double sn = Sh[i];
            const int en = M.epoch_of(sn);
            const double Hn = M.Hc[en] + (sn - M.T[en]) * M.inv2N[en];
            const double need = (Hn - Hc) * kd;
            if (!(g.ebuf > need)) break;
            g.ebuf -= need;
            Hc = Hn; lower = sn; ++i;
        }
        const double C = Hc + g.ebuf / kd;
        int es = 0;
        while (es + 1 < M.E && M.Hc[es + 1] <= C) ++es;
        double tc = M.T[es] + (C - M.Hc[es]) / M.inv2N[es];
        if (tc < lower) tc = lower;
        if (i < ns && tc > Sh[i]) tc = Sh[i];
        g.ebuf = -smc_log(uni(slot));
        if (rec_p && !M.vb_coal.empty()) upd_fac *= M.vb_coal[es];   /* particle.cpp:266-272 */
        /* One coal-opportunity record per interval between h and the event (record_all_event, particle.cpp:251-300) */
        if (rec_p && record_events) {
            double t = h;
            int e = e0;
            i = i0;
            for (;;) {
                double tn_node = i < ns ? Sh[i] : HUGE_VAL;
                double tn_ep = M.epoch_end(e);
                double tn = std::min(tn_node, tn_ep);
                int k = i < ns ? nl - i : 1;
                const bool last = !(tn < tc);
                const double t1 = last ? tc : tn;
                if ((M.recflags[e] & REC_COALMIGR) && e <= limit) {
                    Ev* ev = new_event(*rec_p, e, 1, t, t1, x, x, k);
                    if (last) ev->event = 1;
                }
                if (last) break;
                t = tn;
                if (tn_node <= tn) ++i;
                if (tn_ep <= tn) ++e;
            }
        }
        return tc;
    }

    /* Forest::sampleNextBase via ForestState::sampleNextBase (particle.cpp:1195-1254), multiplicity 1 */
    void sample_next_base(int64_t slot, Particle& p, double x) {
        SlotRng& g = rng[slot];
        /* with a guide the sampling rate is that of the particle's current segment and the draw is limited to the
         * segment (sampleExpoLimit(rate, distance_until_rate_change), particle.cpp:1203-1232) */
        const int K = (int)M.seg_pos.size();
        const bool use_guide = M.guided && stream == 0;          /* calibration uses the true rate (particle.cpp:1211-1218) */
        double rho_here = use_guide ? M.seg_rho[p.ridx] : M.rho;
        double seg_end = (use_guide && p.ridx + 1 < K && M.seg_pos[p.ridx + 1] < M.L) ? M.seg_pos[p.ridx + 1] : M.L;
        double rate = rho_here * p.Ltree;
        double limit = seg_end - x;
        double need = limit * rate;
        if (g.ebuf > need) {
            g.ebuf -= need;
            p.next_base = seg_end;
        } else {
            double nb = x + g.ebuf / rate;
            g.ebuf = -smc_log(uni(slot));
            if (nb == x) nb = std::nextafter(x, x * 2 + 1);   /* particle.cpp:1238-1244 */
            if (nb > seg_end) nb = seg_end;
            p.next_base = nb;
        }
    }

    /* Forest::buildInitialTree(true) [reconstructed]: add the samples one at a time, each new
     * leaf coalescing into the partial tree; coalescences are recorded at position 0. */
    void build_initial_tree(int64_t slot, Particle& p) {
        if (M.P > 1) { mp_build_initial_tree(slot, p); return; }
        const int n = M.n;
        Tree& t = p.tr;
        int root = 0;
        for (int i = 1; i < n; ++i) {
            int ni = i - 1;
            double tc = coalesce_up(slot, &p, t.S, ni, i, 0.0, 0.0, M.E - 1);
            apply_vb(p);
            int pr = -1, ps = 0;
            int k = lineages_at(t, ni, tc, -1, &pr, &ps);
            bool above_root = (ni == 0) || (tc >= t.S[ni - 1]);
            int kk = above_root ? 1 : k;
            double u = uni(slot);
            int idx = std::min((int)(u * (double)kk), kk - 1);
            if (above_root) {
                if (record_trees) push_tree_event(p, 1, 0.0, tc, (2u << i) - 1u);
                insert_node(t, ni, tc, i, -1, 0, root);
            } else {
                lineages_at(t, ni, tc, idx, &pr, &ps);
                if (record_trees) push_tree_event(p, 1, 0.0, tc, (1u << i) | desc_mask(t, t.C[pr][ps]));
                insert_node(t, ni, tc, i, pr, ps, root);
            }
            root = n + ni;     /* the top-ranked node is the root of the partial tree */
        }
        p.Ltree = tree_length(t, n);
    }

    /* One genealogy update at position x: samplePoint (particle.cpp:1060-1126, unbiased case),
     * cut, coalesce, re-attach.  Returns the recombination height through *h_out. */
    void genealogy_update(int64_t slot, Particle& p, double x, int limit, double* h_out) {
        const int n = M.n;
        Tree& t = p.tr;
        if (M.P == 1) prefetch_update_uniforms(slot);      /* dropped after the sample_next_base that follows the update */
        /* --- sample the recombination point on the (possibly height-weighted) local tree (one uniform) --- */
        double prev = 0.0, h = 0.0;
        int lin = 0, slice = 0;
        last_iw = 1.0;
        last_rbiw = 1.0;
        int g_rp = -1, g_sb = 0;
        if (M.guided && stream == 0) {
            /* samplePoint with a recombination guide (particle.cpp:942-1126): every branch carries a relative rate --
             * leaf: the guide's rate for that sample in the current segment; binary node: the arithmetic mean of its
             * children (:1015); the two branches below the root: the larger of the two (:958, 1091) -- times the
             * strength of its height band.  Pieces are visited branch by branch in slot order, bands ascending. */
            const int nbands = (int)M.bias_S.size();
            double brate[2 * NMAX];
            for (int i = 0; i < n; ++i) brate[i] = M.leaf_rate[(size_t)p.ridx * n + i];
            for (int r = 0; r < n - 1; ++r) brate[n + r] = (brate[t.C[r][0]] + brate[t.C[r][1]]) * 0.5;
            const double rroot = std::max(brate[t.C[n - 2][0]], brate[t.C[n - 2][1]]);
            auto visit = [&](auto&& fn) {
                for (int r = 0; r < n - 1; ++r)
                    for (int sdx = 0; sdx < 2; ++sdx) {
                        int c = t.C[r][sdx];
                        double rb = (r == n - 2) ? rroot : brate[c];
                        double lo_b = node_h(t, c), hi_b = t.S[r];
                        for (int b = 0; b < nbands; ++b) {
                            double lo_ = std::max(lo_b, M.bias_H[b]);
                            double hi_ = std::min(hi_b, M.bias_H[b + 1]);
                            if (hi_ > lo_) { if (fn(r, sdx, lo_, hi_, rb * M.bias_S[b])) return; }
                            if (M.bias_H[b + 1] >= hi_b) break;
                        }
                    }
            };
            double Lw = 0.0;
            visit([&](int, int, double lo_, double hi_, double wt) { Lw += wt * (hi_ - lo_); return false; });
            double rr = uni(slot) * Lw;
            double l_lo = 0, l_hi = 0, l_wt = 1;
            bool sel = false;
            visit([&](int r, int sdx, double lo_, double hi_, double wt) {
                double wlen = wt * (hi_ - lo_);
                l_lo = lo_; l_hi = hi_; l_wt = wt; g_rp = r; g_sb = sdx;
                if (rr < wlen) { sel = true; return true; }
                rr -= wlen;
                return false;
            });
            (void)sel;
            h = l_lo + rr / l_wt;
            if (!(h < l_hi)) h = l_lo;
            if (h < l_lo) h = l_lo;
            double sampled = l_wt / Lw;
            double target = 1.0 / p.Ltree;
            last_iw = target / sampled;
            /* the position itself was drawn at the guide's rate: density ratio of the event, true over guide rate.  (The
             * reference returns samplePoint's weight through the absent Forest::sampleNextGenealogy; without this factor
             * the guided sampler is not an importance sampler of the model -- the no-data test checks E[w] = 1.) */
            last_iw *= M.rho / M.seg_rho[p.ridx];
            if (M.biased) {
                /* importance weight of the height bias alone (particle.cpp:1113-1121) */
                double Lrw = 0.0, pv = 0.0;
                for (int ri = 0; ri < n - 1; ++ri) {
                    int k = n - ri;
                    double top = t.S[ri];
                    for (int b = 0; b < nbands; ++b) {
                        double lo_ = std::max(pv, M.bias_H[b]);
                        double hi_ = std::min(top, M.bias_H[b + 1]);
                        if (hi_ > lo_) Lrw += ((double)k * M.bias_S[b]) * (hi_ - lo_);
                        if (M.bias_H[b + 1] >= top) break;
                    }
                    pv = top;
                }
                int idx = 0;
                while (idx + 1 < nbands && M.bias_H[idx + 1] < h) ++idx;
                double recomb_density = M.bias_S[idx] / Lrw;
                last_rbiw = target / recomb_density;
            }
        } else if (!M.biased) {
            double r = uni(slot) * p.Ltree;
            for (int ri = 0; ri < n - 1; ++ri) {
                int k = n - ri;
                double d = t.S[ri] - prev;
                double seg = (double)k * d;
                if (r < seg || ri == n - 2) {
                    double q = r / d;
                    lin = std::min((int)q, k - 1);
                    h = prev + (q - (double)lin) * d;
                    if (!(h < t.S[ri])) h = prev;
                    slice = ri;
                    break;
                }
                r -= seg;
                prev = t.S[ri];
            }
        } else {
            /* samplePoint / accumulateBranchLengths (particle.cpp:1020-1126): branch length in height band b
             * counts bias_S[b]-fold.  Pieces = (time slice) x (band), ascending in height. */
            const int nb = (int)M.bias_S.size();
            double Lw = 0.0;
            {
                double pv = 0.0;
                int b = 0;
                for (int ri = 0; ri < n - 1; ++ri) {
                    int k = n - ri;
                    double top = t.S[ri];
                    while (b + 1 < nb && M.bias_H[b + 1] <= pv) ++b;
                    int bb = b;
                    for (;;) {
                        double lo_ = std::max(pv, M.bias_H[bb]);
                        double hi_ = std::min(top, M.bias_H[bb + 1]);
                        if (hi_ > lo_) Lw += ((double)k * M.bias_S[bb]) * (hi_ - lo_);
                        if (M.bias_H[bb + 1] >= top || bb + 1 >= nb) break;
                        ++bb;
                    }
                    pv = top;
                }
            }
            double r = uni(slot) * Lw;
            double wloc = 1.0;
            double l_lo = 0, l_hi = 0, l_str = 1; int l_k = 1;   /* last piece (fallback against rounding) */
            bool sel = false;
            double pv = 0.0;
            int b = 0;
            for (int ri = 0; ri < n - 1 && !sel; ++ri) {
                int k = n - ri;
                double top = t.S[ri];
                while (b + 1 < nb && M.bias_H[b + 1] <= pv) ++b;
                int bb = b;
                for (;;) {
                    double lo_ = std::max(pv, M.bias_H[bb]);
                    double hi_ = std::min(top, M.bias_H[bb + 1]);
                    if (hi_ > lo_) {
                        double wlen = ((double)k * M.bias_S[bb]) * (hi_ - lo_);
                        l_lo = lo_; l_hi = hi_; l_str = M.bias_S[bb]; l_k = k; slice = ri;
                        if (r < wlen) { sel = true; break; }
                        r -= wlen;
                    }
                    if (M.bias_H[bb + 1] >= top || bb + 1 >= nb) break;
                    ++bb;
                }
                pv = top;
            }
            {
                double q = r / (l_str * (l_hi - l_lo));
                lin = std::min((int)q, l_k - 1);
                if (lin < 0) lin = 0;
                h = l_lo + (q - (double)lin) * (l_hi - l_lo);
                if (!(h < l_hi)) h = l_lo;
                wloc = l_str;
            }
            /* importance weight (particle.cpp:1106-1108): target density 1/L over sampled density w/Lw */
            double sampled = wloc / Lw;
            double target = 1.0 / p.Ltree;
            last_iw = target / sampled;
            last_rbiw = last_iw;
        }
        (void)slice;
        int rp = 0, sb = 0;
        if (g_rp >= 0) { rp = g_rp; sb = g_sb; }
        else lineages_at(t, n - 1, h, lin, &rp, &sb);   /* branch b = slot (rp,sb); its parent p has rank rp */
        *h_out = h;
        {   /* get_descendants (descendants.hpp:22-33) of the cut branch, on the tree before it changes */
            uint64_t below[2 * NMAX];
            for (int i = 0; i < n; ++i) below[i] = 1ull << i;
            for (int r = 0; r < n - 1; ++r) below[n + r] = below[t.C[r][0]] | below[t.C[r][1]];
            last_desc = below[t.C[rp][sb]];
        }
        if (M.P > 1) { mp_genealogy_rest(slot, p, x, limit, rp, sb, h); return; }
        /* --- coalesce upwards against the full old tree (SMC': the cut branch's stub is a target) --- */
        double Sold[NMAX - 1];
        for (int i = 0; i < n - 1; ++i) Sold[i] = t.S[i];
        double tc = coalesce_up(slot, &p, Sold, n - 1, n, h, x, limit);
        last_tc = tc;
        last_first_event = tc;
        double Sp = t.S[rp];
        /* --- detach: remove p, sibling takes its place --- */
        int b_id = t.C[rp][sb], s_id = t.C[rp][1 - sb];
        bool p_was_root = (rp == n - 2);
        remove_rank(t, n - 1, rp, s_id, &b_id, &s_id);
        int ni = n - 2;
        int troot = p_was_root ? s_id : n + (ni - 1);
        /* --- target lineage: slots of the pruned tree crossing tc, then ROOT, then STUB --- */
        int pr = -1, ps = 0;
        int nslots = lineages_at(t, ni, tc, -1, &pr, &ps);
        bool has_root = tc >= node_h(t, troot);
        bool has_stub = tc < Sp;
        int k = nslots + (has_root ? 1 : 0) + (has_stub ? 1 : 0);
        double u = uni(slot);
        int idx = std::min((int)(u * (double)k), k - 1);
        last_sp = Sp;
        last_changed = !(has_stub && idx == k - 1);
        if (record_trees) {
            /* R: the cut (record_recomb_event, particle.cpp:379-389); C: the floating lineage's coalescence
             * (particle.cpp:292-298), with the samples below the node it creates -- its own only when it goes back
             * into its branch (what smcsmc/trees2tskit.py:161-170 calls a back coalescence) */
            const uint32_t cut = (uint32_t)last_desc;
            uint32_t dn = cut;
            if (idx < nslots) {
                int qr = -1, qs = 0;
                lineages_at(t, ni, tc, idx, &qr, &qs);
                dn = cut | desc_mask(t, t.C[qr][qs]);
            } else if (has_root && idx == nslots) {
                dn = (1u << n) - 1u;
            }
            /* the coalescence is recorded while the new genealogy is sampled, the recombination afterwards
             * (record_recomb_event follows sampleNextGenealogy, particle.cpp:861-903): the list then reads R, C */
            push_tree_event(p, 1, x, tc, dn);
            push_tree_event(p, 0, x, h, cut);
        }
        if (idx < nslots) {
            lineages_at(t, ni, tc, idx, &pr, &ps);
            insert_node(t, ni, tc, b_id, pr, ps, troot);
        } else if (has_root && idx == nslots) {
            insert_node(t, ni, tc, b_id, -1, 0, troot);
        } else {
            /* coalesced back into its own branch above the cut: tree unchanged -> restore p */
            if (p_was_root) {
                insert_node(t, ni, Sp, b_id, -1, 0, troot);
            } else {
                /* the sibling lineage's slot at time Sp */
                int want = -1, c = 0;
                int R = 0;
                while (R < ni && t.S[R] <= Sp) ++R;
                for (int rr = R; rr < ni && want < 0; ++rr)
                    for (int s = 0; s < 2 && want < 0; ++s) {
                        int id = t.C[rr][s];
                        if (id < n || id - n < R) {
                            if (id == s_id) want = c;
                            ++c;
                        }
                    }
                lineages_at(t, ni, Sp, want, &pr, &ps);
                insert_node(t, ni, Sp, b_id, pr, ps, troot);
            }
        }
        p.Ltree = tree_length(t, n);
        ++n_recomb;
    }

    /* record_recomb_event (particle.cpp:360-390): mark the rectangle of the new stretch whose
     * time range contains h */
    void record_recomb_event(Particle& p, double h, int limit) {
        int e = M.epoch_of(h);
        if (e > limit) return;
        if (!(M.recflags[e] & REC_RECOMB)) return;
        for (Ev* ev = p.head[e]; ev; ev = ev->parent) {
            if (ev->kind == 0 && ev->t0 <= h && h <= ev->t1) {
                if (ev->x0 == p.x_mark && !ev->event) { ev->event = 1; ev->ev_t = h; ev->desc = last_desc; }
                return;
            }
        }
    }

    /* trackLocalTreeBranchLength (particle.cpp:699-730) on the rank-sorted tree */
    double tracked_length(const Tree& t, const int8_t* data) const {
        const int n = M.n;
        double stbl[2 * NMAX];
        for (int i = 0; i < n; ++i) stbl[i] = data[i] >= 0 ? 0.0 : -1.0;
        double total = 0.0;
        for (int r = 0; r < n - 1; ++r) {
            int c0 = t.C[r][0], c1 = t.C[r][1];
            double l = stbl[c0], rr = stbl[c1];
            if (l >= 0.0) l += t.S[r] - node_h(t, c0);
            if (rr >= 0.0) rr += t.S[r] - node_h(t, c1);
            double v;
            if (l >= 0.0 && rr >= 0.0) { total = l + rr; v = total; }
            else if (l >= 0.0) v = l;
            else v = rr;
            stbl[n + r] = v;
        }
        return total;
    }

    /* calculate_likelihood + cal_partial_likelihood_infinite (particle.cpp:625-680) */
    double site_likelihood(const Tree& t, const int* hap) const { return site_likelihood_aa(t, hap, M.ancestral_aware); }
    double site_likelihood_aa(const Tree& t, const int* hap, bool ancestral_aware) const {
        const int n = M.n;
        double m0[2 * NMAX], m1[2 * NMAX];
        for (int i = 0; i < n; ++i) {
            m0[i] = hap[i] == 1 ? 0.0 : 1.0;
            m1[i] = hap[i] == 0 ? 0.0 : 1.0;
        }
        for (int r = 0; r < n - 1; ++r) {
            int c0 = t.C[r][0], c1 = t.C[r][1];
            double tl = t.S[r] - node_h(t, c0);
            double trr = t.S[r] - node_h(t, c1);
            double pl = fastexp(-tl * M.mu);
            double pr = fastexp(-trr * M.mu);
            m0[n + r] = (m0[c0] * pl + m1[c0] * (1 - pl)) * (m0[c1] * pr + m1[c1] * (1 - pr));
            m1[n + r] = (m1[c0] * pl + m0[c0] * (1 - pl)) * (m1[c1] * pr + m0[c1] * (1 - pr));
        }
        int root = n + n - 2;
        double p0 = ancestral_aware ? 1.0 : 0.5, p1 = ancestral_aware ? 0.0 : 0.5;
        return m0[root] * p0 + m1[root] * p1;
    }

    /* ---------------- auxiliary particle filter (particle.cpp:439-617) ---------------- */
    /* height of the first local node above leaf i: its coalescent parent, or the first migration on its branch */
    double leaf_parent_height(const Tree& t, int leaf, int* parent_rank) const {
        const int n = M.n;
        int pr = -1;
        for (int r = 0; r < n - 1 && pr < 0; ++r)
            if (t.C[r][0] == leaf || t.C[r][1] == leaf) pr = r;
        *parent_rank = pr;
        double hgt = t.S[pr];
        if (M.P > 1)
            for (int m = 0; m < t.nm; ++m)
                if (t.Mb[m] == leaf) { hgt = t.Mt[m]; pr = -2 - m; *parent_rank = pr; break; }   /* a migrating node */
        return hgt;
    }

    double lookahead_likelihood(const Tree& t, double Ltree, int64_t row) const {
        const int n = M.n, Q = la_Q, D = la_D;
        const double* fsd = &la_fsd[row * n];
        const double* rmr = &la_rmr[row * n];
        const int8_t* unph = &la_unph[row * n];
        const double recomb_rate = M.rho, mut_rate = M.mu;
        const double rel_rho[2] = {1.0, 0.5}, rel_rho_p[2] = {0.5, 0.5};
        double likelihood = 1.0;
        double rho_tbl = 2 * recomb_rate * (n - 1) / n;
        double mut_prob[NMAX];
        double lh[NMAX]; int lpar[NMAX];
        for (int i = 0; i < n; ++i) { mut_prob[i] = 0; lh[i] = leaf_parent_height(t, i, &lpar[i]); }
        for (int i = 0; i < n; i++) {
            double p = 0;
            double si = fsd[i];
            double li = lh[i];
            if (unph[i]) li += lh[i + 1];
            double rel_mut_rate = rmr[i];
            double li_mu = li * mut_rate * rel_mut_rate;
            mut_prob[i] = li_mu;
            if (unph[i]) mut_prob[i + 1] = li_mu;
            for (int r = 0; r < 2; r++) {
                double li_rho = li * rho_tbl * rel_rho[r];
                double fe = fastexp_approx(-(li_rho + li_mu) * std::fabs(si));
                for (int q = 0; q < Q; ++q) {
                    double qbot = (q == 0 ? 0.0 : la_q[q - 1]);
                    double qtop = (q == Q - 1 ? 1.0 : la_q[q]);
                    double l_prime = la_tbl[i * Q + q];
                    double lprime_mu = l_prime * mut_rate * rel_mut_rate;
                    double div = (li_rho + li_mu - lprime_mu);
                    if (std::fabs(div) < (li_rho + li_mu + lprime_mu) * 1e-5) lprime_mu = lprime_mu * 1.0001;
                    if (si > 0) {
                        p += rel_rho_p[r] * (qtop - qbot) * ((li_rho * lprime_mu * fastexp_approx(-lprime_mu * si) +
                                                              (li_mu - lprime_mu) * (li_rho + li_mu) * fe)
                                                             / (li_rho + li_mu - lprime_mu));
                    } else {
                        p += rel_rho_p[r] * (qtop - qbot) * ((li_rho * fastexp_approx(-lprime_mu * (-si)) +
                                                              (li_mu - lprime_mu) * fe)
                                                             / (li_rho + li_mu - lprime_mu));
                    }
                }
            }
            likelihood *= p;
            if (unph[i]) i++;        /* do not double-count unphased singletons */
        }
        if (apf >= 2) {
            double l_mean = 0.0;
            for (int i = 0; i < n; i++) l_mean += la_tbl[i * Q + (Q - 1)] / n;
            double rho_c = 4 * recomb_rate * (n - 2) / n;
            double rhoprime_c = recomb_rate * (n - 1);
            double p_equilibrium = 2.0 / (3 * (n - 1));
            const int nd = la_nd[row];
            for (int k = 0; k < nd; ++k) {
                const int8_t* di = &la_didx[(row * D + k) * 4];
                const double fed = la_ddist[(row * D + k) * 2], led = la_ddist[(row * D + k) * 2 + 1];
                int ph1, ph2;
                for (ph1 = 0; ph1 <= di[2]; ph1++) {
                    for (ph2 = 0; ph2 <= di[3]; ph2++) {
                        int a = di[0] + ph1, b = di[1] + ph2;
                        if (lpar[a] >= 0 && lpar[a] == lpar[b]) {
                            double l = lh[a];
                            double p = 0;
                            for (int r = 0; r < 2; r++) {
                                double exp_rho = fastexp_approx(-rho_c * rel_rho[r] * l * led);
                                p += rel_rho_p[r] * exp_rho + p_equilibrium * (1.0 - exp_rho);
                            }
                            likelihood *= p;
                            ph1 = ph2 = 99;
                        }
                    }
                }
                if (ph1 < 99) {
                    double mutprob = (mut_prob[di[0]] + mut_prob[di[1]]) * 0.5;
                    double p = 0;
                    for (int r = 0; r < 2; r++)
                        p += rel_rho_p[r] * (mutprob + (1.0 - mutprob) * p_equilibrium *
                                             (1.0 - fastexp_approx(-rhoprime_c * rel_rho[r] * l_mean * fed)));
                    likelihood *= p;
                }
            }
        }
        if (la_split[row] > -1 && apf >= 3) {
            double rate_of_change = Ltree * recomb_rate / 2;
            double p_nochange = fastexp_approx(-rate_of_change * la_split[row]);
            int hap[NMAX];
            for (int i = 0; i < n; ++i) hap[i] = la_salleles[row * n + i];
            double p_splitdata = site_likelihood_aa(t, hap, false);
            int k = la_sk[row];
            double p_correct_split = k / double(4.0 * n * n);
            if (apf == 4) {
                double nCk = 1.0;
                for (int i = 1; i <= k; i++) nCk *= (n - i + 1) / double(i);
                p_correct_split = 1.0 / nCk;
            }
            double etbl = la_mean_tbl;
            double split_branch_length = k * etbl / (2 * n * (0.577 * smc_log((double)n)));
            double p = p_nochange * p_splitdata + (1.0 - p_nochange) * p_correct_split * mut_rate * split_branch_length;
            likelihood *= p;
        }
        return likelihood;
    }

    /* ---------------- ParticleContainer ---------------- */

    /* particleContainer.cpp:33-65 */
    void init_prior(double initial_position) {
        parts.assign(Np, Particle());
        rng.assign(Np, SlotRng{0, 0.0});
        for (int64_t i = 0; i < Np; ++i) {
            Particle& p = parts[i];
            p.head.assign(M.E, nullptr);
            rng[i].ebuf = -smc_log(uni(i));
            p.w_post = 1.0 / (double)Np;
            p.w_pilot = 1.0 / (double)Np;
            build_initial_tree(i, p);
            sample_next_base(i, p, 0.0);
            open_stretch(p, 0.0, M.E - 1);
        }
        cur_pos = initial_position;
        logl = 0;
        const int E = M.E;
        const int P = M.P;
        coal_count.assign(E * P, 0); coal_opp.assign(E * P, 0); coal_w2.assign(E * P, 0);
        mig_count.assign(E * P * P, 0); mig_opp.assign(E * P, 0); mig_w2.assign(E * P, 0);
        if (local_map) {
            size_t nb = (size_t)(M.L / 100.0) + 4;
            local_opp.assign(nb, 0.0);
            local_cnt.assign(M.n + 2, std::vector<double>(nb, 0.0));
        }
        rec_count.assign(E, 0); rec_opp.assign(E, 0); rec_w2.assign(E, 0);
        counted_to.assign(E, 0);
    }

    /* ForestState::extend_ARG (particle.cpp:743-918), unbiased sampling, multiplicity 1 */
    void extend_particle(int64_t slot, double extend_to, int leaf_status, const int8_t* data, int limit) {
        Particle& p = parts[slot];
        double updated_to = cur_pos;
        double B;
        switch (leaf_status) {
            case -1: B = 0; break;
            case 1: B = p.Ltree; break;
            default: B = tracked_length(p.tr, data); break;
        }
        while (updated_to < extend_to) {
            double new_updated_to = std::min(extend_to, p.next_base);
            double f = fastexp(-M.mu * B * (new_updated_to - updated_to));
            p.w_post *= f;
            p.w_pilot *= f;
            if (M.guided) {
                /* importance_weight_over_segment (particle.cpp:1138-1181): true rate over guide rate for the stretch
                 * without recombination.  (The reference gates it on model().biased_sampling, particle.cpp:811-813, a
                 * flag of the absent fork; a guide without it would bias every estimate, so it is applied with any guide.) */
                double dist = new_updated_to - updated_to;
                double target_rate = dist * M.rho * p.Ltree;
                double sampled_rate = dist * M.seg_rho[p.ridx] * p.Ltree;
                double iws = fastexp(sampled_rate - target_rate);
                p.w_post *= iws;
                p.w_pilot *= iws;
            }
            updated_to = new_updated_to;
            if (updated_to < extend_to) {
                if (M.guided && p.ridx + 1 < (int)M.seg_pos.size() && updated_to == M.seg_pos[p.ridx + 1]) {
                    /* reached a change of the guide rate: no genealogy change, new draw under the new rate
                     * (particle.cpp:822-826); the open stretch simply continues (D6) */
                    p.ridx += 1;
                    sample_next_base(slot, p, updated_to);
                    continue;
                }
                /* a recombination has occurred (particle.cpp:828-903) */
                close_stretch(p, updated_to);
                double h;
                genealogy_update(slot, p, updated_to, limit, &h);
                apply_vb(p);
                if (leaf_status == 0) B = tracked_length(p.tr, data);
                if (leaf_status == 1) B = p.Ltree;
                if (M.biased || M.guided) {
                    /* particle.cpp:866-891: immediate vs delayed application of the importance weight */
                    double iw = last_iw;
                    double rbiw = last_rbiw;                       /* without a guide both weights coincide */
                    /* RESAMPLE_DELAY_RECOMB: the cut height; _COAL: first_coal_height_; _COALMIGR: first_event_height_ */
                    const int dtype = M.delay_type & 3;
                    double delay_height = dtype == 0 ? h : (dtype == 2 ? last_first_event : last_tc);
                    int idx = 0;
                    while (idx + 1 < (int)M.bias_H.size() && M.bias_H[idx + 1] < delay_height) ++idx;
                    if (idx >= (int)M.bias_S.size()) idx = (int)M.bias_S.size() - 1;
                    /* delay_type bit 2 (an experiment switch of the oracle, not a reference option): every factor goes
                     * through the delayed store, as on the branch the two-population bands were calibrated on
                     * ("2b3a_wo_apply_immediately_hack", test_two_pops.py:50) */
                    if (M.bias_S[idx] == 1.0 && !(M.delay_type & 4)) {
                        p.w_post *= rbiw; p.w_pilot *= rbiw;
                        iw /= rbiw;
                    }
                    double delay = M.app_delays[M.epoch_of(delay_height)];   /* find_delay, particle.cpp:733-740 */
                    adjust_with_delay(p, iw, delay, updated_to);
                }
                sample_next_base(slot, p, updated_to);
                drop_update_uniforms();
                open_stretch(p, updated_to, limit);
                record_recomb_event(p, h, limit);
            }
        }
        if (M.biased || M.guided) apply_due(p, extend_to);
    }

    /* adjustWeightsWithDelay(adjustment, delay, k = 3) (particle.hpp:189-198) with DelayedFactor (59-77):
     * the factor is applied to the pilot weight in three equal parts at cur+d/7, cur+3d/7, cur+d */
    void adjust_with_delay(Particle& p, double adj, double delay, double cur) {
        p.w_post *= adj;
        if ((adj > 0.99 && adj < 1.01) || (delay <= 1)) {
            p.w_pilot *= adj;
            return;
        }
        /* the store of the device path is bounded (delay_cap entries per particle): a full store stops the run, as every
         * other bounded ring does -- or, with delay_evict, the earliest pending factor is applied ahead of its position to
         * make room, and the event is counted */
        if (p.dcount == delay_cap) {
            if (!delay_evict) throw std::runtime_error("delayed-factor store overflow");
            while (p.dcount == delay_cap) { apply_earliest(p); ++n_delay_evict; }
        }
        p.total_delayed *= adj;
        double final_pos = cur + delay;
        double delta = (final_pos - cur) / 7.0;
        int i = p.dcount++;
        if ((int)p.dpos.size() <= i) { p.dpos.resize(i + 1); p.dfac.resize(i + 1); p.ddelta.resize(i + 1); p.dk.resize(i + 1); }
        p.dpos[i] = cur + delta;
        p.dfac[i] = smc_exp(smc_log(adj) * (1.0 / 3));   /* pow(factor, 1.0/k), libm-free */
        p.ddelta[i] = delta;
        p.dk[i] = 3;
        if (p.dcount > delay_peak) delay_peak = p.dcount;
    }
    /* applyDelayedAdjustment (particle.hpp:199-209); the entry with the smallest position (ties: lowest index) */
    void apply_earliest(Particle& p) {
        int m = 0;
        for (int i = 1; i < p.dcount; ++i) if (p.dpos[i] < p.dpos[m]) m = i;
        p.w_pilot *= p.dfac[m];
        p.total_delayed /= p.dfac[m];
        if (p.dk[m] > 1) {
            p.dpos[m] = p.dpos[m] + 2 * p.ddelta[m];
            p.ddelta[m] = 2 * p.ddelta[m];
            p.dk[m] -= 1;
        } else {
            int last = --p.dcount;
            p.dpos[m] = p.dpos[last]; p.dfac[m] = p.dfac[last]; p.ddelta[m] = p.ddelta[last]; p.dk[m] = p.dk[last];
        }
    }
    void apply_due(Particle& p, double extend_to) {      /* particle.cpp:910-916 */
        for (;;) {
            if (p.dcount == 0) return;
            int m = 0;
            for (int i = 1; i < p.dcount; ++i) if (p.dpos[i] < p.dpos[m]) m = i;
            if (!(p.dpos[m] < extend_to)) return;
            apply_earliest(p);
        }
    }

    /* update_state_to_data (particleContainer.cpp:441-466) */
    void update_segment(const smco_segments* sg, int64_t s) {
        const int n = M.n;
        const int8_t* data = sg->alleles + s * n;
        double seg_end = sg->start[s] + sg->length[s];
        double extend_to = std::min(seg_end, M.L);
        int limit = sg->max_record_epoch[s];
        cur_limit = limit;
        /* extend_ARGs (particleContainer.cpp:98-135) */
        int missing = 0;
        for (int i = 0; i < n; ++i) missing += data[i] == -1;
        int leaf_status = 0;
        if (missing == 0) leaf_status = 1;
        if (missing == n) leaf_status = -1;
        for (int64_t i = 0; i < Np; ++i) extend_particle(i, extend_to, leaf_status, data, limit);
        cur_pos = std::max(cur_pos, extend_to);
        /* update_weight_at_site (particleContainer.cpp:187-224) */
        if (sg->state[s] == 0) {
            int hap[NMAX];
            int ncfg = 1;
            /* calculate_initial_haplotype_configuration (pc.cpp:138-160) */
            for (int i = 0; i < n; ++i) hap[i] = data[i];
            for (int i = 0; i + 1 < n; i += 2) {
                bool het = (data[i] == 2) || (M.dephase && data[i] + data[i + 1] == 1);
                if (het) { ncfg *= 2; hap[i] = 0; hap[i + 1] = 1; }
            }
            double norm = 1.0 / ncfg;
            std::vector<int> h0(hap, hap + n);
            for (int64_t pi = 0; pi < Np; ++pi) {
                Particle& p = parts[pi];
                for (int i = 0; i < n; ++i) hap[i] = h0[i];
                double lik = 0;
                for (;;) {
                    lik += site_likelihood(p.tr, hap);
                    if (ncfg == 1) break;
                    /* next_haplotype (pc.cpp:163-181) */
                    bool more = false;
                    for (int i = 0; i + 1 < n; i += 2) {
                        bool het = (data[i] == 2) || (M.dephase && data[i] + data[i + 1] == 1);
                        if (!het) continue;
                        if (hap[i] == 0) { hap[i] = 1; hap[i + 1] = 0; more = true; break; }
                        hap[i] = 0; hap[i + 1] = 1;
                    }
                    if (!more) break;
                }
                lik *= norm;
                p.w_post *= lik;
                p.w_pilot *= lik;
            }
        }
        /* update_lookahead_likelihood (pc.cpp:227-240): remove the previous look-ahead factor from the pilot weight,
         * include the new one.  (The reference does this between the extension and the site weight, pc.cpp:455-459;
         * the pilot weight is the same product of factors either way.) */
        if (apf > 0) {
            for (int64_t pi = 0; pi < Np; ++pi) {
                Particle& p = parts[pi];
                p.w_pilot /= p.lookahead;
                p.lookahead = 1.0;
                double lk = lookahead_likelihood(p.tr, p.Ltree, s);
                p.lookahead *= lk;
                p.w_pilot *= lk;
            }
        }
        normalize();
    }

    /* normalize_probability (particleContainer.cpp:420-438), canonical sum (D3) */
    double last_T = 1, last_inv = 1;
    std::vector<double> raw_pilot;   /* pilot weights before normalisation: ESS / scan operate on these (D5) */
    void normalize() {
        std::vector<double> w(Np);
        raw_pilot.resize(Np);
        for (int64_t i = 0; i < Np; ++i) { w[i] = parts[i].w_post; raw_pilot[i] = parts[i].w_pilot; }
        double T = canon_sum(w.data(), Np);
        if (!(T > 0)) throw std::runtime_error("Zero or negative probabilities");
        logl += smc_log(T);
        double inv = 1.0 / T;
        for (int64_t i = 0; i < Np; ++i) { parts[i].w_post *= inv; parts[i].w_pilot *= inv; }
        last_T = T;
        last_inv = inv;
    }

    /* ---------------- CountModel (count.cpp:355-555) ---------------- */
    std::vector<double> update_to;

    void count_single(Ev* ev, double w, int e) {
        /* update_all_counts_single_evolevent (count.cpp:495-555) */
        double x_start = counted_to[e], x_end = update_to[e];
        double ep0 = M.T[e], ep1 = M.epoch_end(e);
        double ts = std::max(0.0, std::min(ep1, ev->t1) - std::max(ep0, ev->t0));
        if (ev->kind == 1) {
            const int P = M.P, ep = e * P + ev->pop;
            if (x_start <= ev->x0 && ev->event == 1) coal_count[ep] += w;
            if (x_start <= ev->x0 && ev->event == 2) mig_count[ep * P + ev->mig_to] += w;
            double opp = ev->weight * ts;                 /* coalevent.hpp:212-214 */
            coal_opp[ep] += w * opp;
            coal_w2[ep] += w * w * opp;
            if (P > 1) {                                  /* coalevent.hpp:215-220, count.cpp:521-526 */
                mig_opp[ep] += w * ts;
                mig_w2[ep] += w * w * ts;
            }
        } else {
            bool end_seq = (M.L == x_end);
            double xs = std::max(0.0, std::min(x_end, ev->x1) - std::max(x_start, ev->x0));
            double opp = ev->weight * ts * xs;            /* coalevent.hpp:224-226 */
            if (ev->event) {                              /* coalevent.hpp:231-235 */
                bool in = (x_start <= ev->x0) && ((ev->x0 < x_end) || end_seq) && (ep0 <= ev->ev_t) && (ev->ev_t < ep1);
                if (in) rec_count[e] += w;
            }
            rec_opp[e] += w * opp;
            rec_w2[e] += w * w * opp;
            if (local_map) {                              /* count.cpp:540-551 */
                double local_x_start = std::max(x_start, ev->x0);
                double local_x_end = std::min(x_end, ev->x1);
                if (local_x_start < local_x_end) {
                    double event_base = ev->event ? ev->x0 : -1.0;       /* recomb_event_base, coalevent.hpp:236-241 */
                    double event_time = -1;
                    uint64_t d = 0;
                    if (x_start <= event_base && event_base <= x_end) { event_time = ev->ev_t; d = ev->desc; }
                    record_local(local_x_start, local_x_end, w, opp, event_base, event_time, d);
                }
            }
        }
    }

    /* record_local_recomb_events (count.cpp:559-613) */
    void record_local(double x_start, double x_end, double weight, double opportunity, double event_base, double event_time,
                      uint64_t descendants) {
        const double iv = 100.0;
        size_t first_index = (size_t)(x_start / iv);
        size_t last_index = (size_t)(1 + x_end / iv);
        double first_interval = std::min((first_index + 1) * iv, x_end) - x_start;
        double last_interval = x_end - std::max((last_index - 1) * iv, x_start);
        double opp_density = weight * opportunity / (x_end - x_start);
        if (last_index >= local_opp.size()) throw std::logic_error("local recombination map too small");
        if (first_index == last_index - 1) {
            local_opp[first_index] += first_interval * opp_density;
            local_opp[first_index + 1] -= first_interval * opp_density;
        } else {
            local_opp[first_index] += first_interval * opp_density;
            local_opp[first_index + 1] += (iv - first_interval) * opp_density;
            local_opp[last_index - 1] += (last_interval - iv) * opp_density;
            local_opp[last_index] -= last_interval * opp_density;
        }
        if (x_start <= event_base && event_base <= x_end) {
            int nd = 0;
            for (int i = 0; i < M.n; ++i) nd += (int)((descendants >> i) & 1);
            size_t index = (size_t)(event_base / iv);
            for (int i = 0; i < M.n; ++i)
                if ((descendants >> i) & 1) local_cnt[i][index] += weight / nd;
            local_cnt[M.n][index] += weight * event_time;
            local_cnt[M.n + 1][index] += weight * smc_log(event_time + 1.0);
        }
    }

    /* update_all_counts (count.cpp:448-491): push posterior weight down one epoch chain */
    void walk_chain(Ev** hp, double w, int e) {
        Ev** pp = hp;
        for (;;) {
            Ev* ev = *pp;
            if (!ev) return;
            if (ev->dead) {           /* fully consumed earlier: unlink (purge_events/remove_event) */
                *pp = nullptr;
                release(ev);
                return;
            }
            ev->acc += w;
            if (++ev->arrived < ev->refs) return;      /* coalevent.hpp:305-309 */
            w = ev->acc;
            ev->acc = 0;
            ev->arrived = 0;
            double start_base = ev->x0;
            if (start_base < update_to[e]) {
                count_single(ev, w, e);
                if (ev->x1 < update_to[e]) ev->dead = 1;
            }
            pp = &ev->parent;
        }
    }

    /* extract_and_update_count (count.cpp:355-415) */
    void count(double current_base, bool end_data) {
        const int E = M.E;
        update_to.assign(E, 0);
        int first = E;
        for (int e = 0; e < E; ++e) {
            double lagging = end_data ? 0 : M.lags[e];
            double x_end = current_base - lagging;
            if ((x_end - counted_to[e]) < lagging * 0.1 && first > e) {
                update_to[e] = counted_to[e];
            } else {
                update_to[e] = x_end;
                first = std::min(first, e);
            }
        }
        for (int64_t i = Np - 1; i >= 0; --i) {
            Particle& p = parts[i];
            if (p.dcount > 0) delayed_count += update_to[E - 1] - counted_to[E - 1];   /* count.cpp:395-397 */
            for (int e = E - 1; e >= first; --e) walk_chain(&p.head[e], p.w_post, e);
        }
        delayed_opp += update_to[E - 1] - counted_to[E - 1];
        for (int e = 0; e < E; ++e) counted_to[e] = update_to[e];
    }

    /* resample (particleContainer.cpp:247-311) + implement_resampling (321-392) */
    int resample(double update_pos) {
        std::vector<double> pil(Np), incl(Np), sq(Np);
        for (int64_t i = 0; i < Np; ++i) { pil[i] = raw_pilot[i]; sq[i] = pil[i] * pil[i]; }
        canon_scan(pil.data(), incl.data(), Np);
        double S1 = incl[Np - 1];
        double S2 = canon_sum(sq.data(), Np);
        double ess = (S1 * S1) / S2;
        tr_ess.push_back(ess);
        double thr = (double)Np * ess_fraction;
        if (!(ess < thr - 1e-6)) { tr_flag.push_back(0); return 0; }
        tr_flag.push_back(1);
        /* systematic_resampling (particleContainer.cpp:474-504) */
        double u = philox_uniform(seed, 0xFFFFFFFFu, 1, (uint64_t)n_resample);
        std::vector<int32_t> lo(Np + 1);
        systematic(incl.data(), Np, u, lo.data());
        /* implement_resampling */
        std::vector<Particle> np_(Np);
        std::vector<int32_t> parents(Np);
        for (int64_t i = 0; i < Np; ++i) {
            int cnt = lo[i + 1] - lo[i];
            Particle& src = parts[i];
            if (cnt == 0) {
                for (Ev* h : src.head) release(h);   /* ~ForestState: particle.cpp:161-187 */
                continue;
            }
            double adj = (S1 * last_inv) / ((double)Np * src.w_pilot);   /* pc.cpp:350-351 */
            src.w_post *= adj;
            src.w_pilot *= adj;
            if (cnt >= 2) close_stretch(src, update_pos);
            for (int32_t q = lo[i]; q < lo[i + 1]; ++q) {
                parents[q] = (int32_t)i;
                Particle& d = np_[q];
                if (q == lo[i] && cnt == 1) { d = std::move(src); continue; }
                d.tr = src.tr; d.w_post = src.w_post; d.w_pilot = src.w_pilot;
                d.next_base = src.next_base; d.Ltree = src.Ltree;
                d.lookahead = src.lookahead; d.ridx = src.ridx;
                d.total_delayed = src.total_delayed; d.dcount = src.dcount;
                d.dpos.assign(src.dpos.begin(), src.dpos.begin() + src.dcount); d.dfac.assign(src.dfac.begin(), src.dfac.begin() + src.dcount);
                d.ddelta.assign(src.ddelta.begin(), src.ddelta.begin() + src.dcount); d.dk.assign(src.dk.begin(), src.dk.begin() + src.dcount);
                d.tree_head = src.tree_head;             /* the pseudo-epoch of tree events is copied with the others */
                d.head = src.head;                       /* copyEventContainers: particle.cpp:139-148 */
                for (Ev* h : d.head) if (h) ++h->refs;
            }
            if (cnt >= 2) for (Ev* h : src.head) release(h);
        }
        parts.swap(np_);
        /* new stretches / fresh recombination positions for copies (pc.cpp:357-368) */
        for (int64_t i = 0; i < Np; ++i) {
            int cnt = lo[i + 1] - lo[i];
            if (cnt < 2) continue;
            for (int32_t q = lo[i]; q < lo[i + 1]; ++q) {
                Particle& d = parts[q];
                int lim = np_[i].mark_limit;
                if (q != lo[i] && update_pos < M.L) sample_next_base(q, d, update_pos);
                open_stretch(d, update_pos, lim);
            }
        }
        if ((int32_t)ev_parents.size() < max_trace_events) {
            ev_seg.push_back((int32_t)seg_done);
            ev_parents.push_back(parents);
        }
        ++n_resample;
        return 1;
    }

    void finish() {
        /* smcsmc.cpp:371-373 */
        normalize();
        count(M.L, true);
    }
};


/* calculate_median_survival_distances (smcsmc.cpp:169-263), batched exactly like the HIP driver:
 * batches of CAL_BATCH independent prior ARGs (Philox stream 2, slot = replicate index) until every
 * epoch has min_events samples or max_trees trees were used; medians + fallbacks as smcsmc.cpp:235-262 */
enum { CAL_BATCH = 16384 };

static void median_survival(const Model& M, uint64_t seed, int min_events, int64_t max_trees, double* median_out,
                            int64_t* trees_used) {
    const int E = M.E, n = M.n;
    std::vector<std::vector<double>> surv(E);
    int64_t trees = 0;
    for (;;) {
        int not_done = 0;
        for (int e = 0; e < E; ++e) not_done += (int)surv[e].size() < min_events;
        if (not_done == 0 || trees >= max_trees) break;
        Filter f;
        f.M = M; f.Np = 1; f.seed = seed; f.stream = 2; f.record_events = false;
        f.rng.assign(1, SlotRng{0, 0.0});
        for (int64_t r = 0; r < CAL_BATCH; ++r) {
            Particle p;
            p.head.assign(E, nullptr);
            /* the replicate's own stream: slot = global replicate index */
            f.rng[0] = SlotRng{0, 0.0};
            f.slot_override = trees + r;
            f.rng[0].ebuf = -smc_log(f.uni(0));
            f.build_initial_tree(0, p);
            double orig[NMAX];
            bool alive[NMAX];
            int nalive = n - 1;
            for (int j = 0; j < n - 1; ++j) { orig[j] = p.tr.S[j]; alive[j] = true; }
            int ep[NMAX];
            for (int j = 0; j < n - 1; ++j) ep[j] = M.epoch_of(orig[j]);
            f.sample_next_base(0, p, 0.0);
            double stop = M.L * 0.6;
            while (nalive > 0 && p.next_base < stop) {
                double x = p.next_base;
                double h;
                f.genealogy_update(0, p, x, -1, &h);
                if (f.last_changed) {
                    for (int j = 0; j < n - 1; ++j)
                        if (alive[j] && orig[j] == f.last_sp) {
                            surv[ep[j]].push_back(x);
                            alive[j] = false;
                            --nalive;
                            break;
                        }
                }
                f.sample_next_base(0, p, x);
                f.drop_update_uniforms();
            }
        }
        trees += CAL_BATCH;
    }
    if (trees_used) *trees_used = trees;
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
}

}  // namespace smco

using namespace smco;

#define GUARD(...)                                                   \
    try { __VA_ARGS__ }                                              \
    catch (const std::exception& e) { g_err = e.what(); return -1; } \
    return 0;

extern "C" {

const char* smco_last_error(void) { return g_err.c_str(); }

static void fill_model(Model& M, const smco_model* m) {
    if (m->n_pops < 1 || m->n_pops > 8) throw std::runtime_error("oracle: n_pops out of range");
    if (m->nsam < 2 || m->nsam > NMAX) throw std::runtime_error("oracle: nsam out of range");
    M.E = m->n_epochs; M.P = m->n_pops; M.n = m->nsam;
    const int E = M.E, P = M.P;
    M.ancestral_aware = m->flags & 1; M.dephase = m->flags & 2;
    M.L = m->loci_length; M.mu = m->mutation_rate; M.rho = m->recombination_rate;
    M.T.assign(m->change_times, m->change_times + E);
    M.inv2N.resize(E);
    for (int e = 0; e < E; ++e) M.inv2N[e] = 1.0 / (2.0 * m->pop_sizes[e * P]);
    M.Hc.assign(E, 0.0);
    for (int e = 0; e + 1 < E; ++e) M.Hc[e + 1] = M.Hc[e] + (M.T[e + 1] - M.T[e]) * M.inv2N[e];
    M.inv2Np.resize(E * P);
    for (int i = 0; i < E * P; ++i) M.inv2Np[i] = 1.0 / (2.0 * m->pop_sizes[i]);
    M.Mrate.assign(E * P * P, 0.0);
    if (m->mig_rates) M.Mrate.assign(m->mig_rates, m->mig_rates + E * P * P);
    M.Mtot.assign(E * P, 0.0);
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < P; ++a) {
            double sum = 0.0;
            for (int b = 0; b < P; ++b) if (b != a) sum += M.Mrate[(e * P + a) * P + b];
            M.Mtot[e * P + a] = sum;
        }
    /* fixed-time moves: deterministic only (single_mig_pop in {0,1}), chains resolved here
     * (dontImplementFixedTimeEvent, particle.cpp:1387-1418) */
    M.jmap.resize(E * P);
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < P; ++a) {
            int cur = a;
            for (int step = 0; step <= P && m->single_mig; ++step) {
                int nxt = cur;
                for (int b = 0; b < P; ++b) {
                    double pr = m->single_mig[(e * P + cur) * P + b];
                    if (pr != 0.0 && pr != 1.0) throw std::runtime_error("oracle: partial single migration events are not supported");
                    if (pr == 1.0 && b != cur) { nxt = b; break; }
                }
                if (nxt == cur) break;
                if (step == P) throw std::logic_error("Cycle detected when moving individuals between populations");
                cur = nxt;
            }
            M.jmap[e * P + a] = cur;
        }
    M.Cc.assign(E * P, 0.0);
    M.Cm.assign(E * P, 0.0);
    for (int e = 0; e + 1 < E; ++e)
        for (int a = 0; a < P; ++a) {
            const double dt = M.T[e + 1] - M.T[e];
            M.Cc[(e + 1) * P + a] = M.Cc[e * P + a] + dt * M.inv2Np[e * P + a];
            M.Cm[(e + 1) * P + a] = M.Cm[e * P + a] + dt * M.Mtot[e * P + a];
        }
    M.Tjoin.assign(E, HUGE_VAL);
    for (int e = E - 2; e >= 0; --e) {
        bool moves = false;
        for (int a = 0; a < P; ++a) moves |= M.jmap[(e + 1) * P + a] != a;
        M.Tjoin[e] = moves ? M.T[e + 1] : M.Tjoin[e + 1];
    }
    if (m->vb_coal_counts) {
        M.vb_coal.resize(E * P);
        for (int i = 0; i < E * P; ++i) M.vb_coal[i] = exp_digamma(m->vb_coal_counts[i]) / m->vb_coal_counts[i];
        M.vb_mig.assign(E * P * P, 1.0);
        if (m->vb_mig_counts)
            for (int i = 0; i < E * P * P; ++i) M.vb_mig[i] = exp_digamma(m->vb_mig_counts[i]) / m->vb_mig_counts[i];
    }
    M.sample_pop.assign(M.n, 0);
    if (m->sample_pops) M.sample_pop.assign(m->sample_pops, m->sample_pops + M.n);
    for (int v : M.sample_pop) if (v < 0 || v >= P) throw std::runtime_error("oracle: sample population out of range");
}

void* smco_create(const smco_model* m, const smco_params* p) {
    try {
        Filter* f = new Filter();
        Model& M = f->M;
        fill_model(M, m);
        if (p->mig_cap > MMAX) throw std::runtime_error("oracle: mig_cap above the storage of the restatement");
        if (p->mig_cap > 0) f->mig_cap = p->mig_cap;
        if (p->delay_cap > 0) f->delay_cap = p->delay_cap;
        f->delay_evict = p->delay_evict != 0;
        M.recflags.assign(m->record_flags, m->record_flags + M.E);
        M.lags.assign(m->lags, m->lags + M.E);
        if (m->n_bias_heights > 0) {
            M.biased = true;
            M.bias_H.push_back(0.0);
            for (int k = 0; k < m->n_bias_heights; ++k) M.bias_H.push_back(m->bias_heights[k]);
            M.bias_H.push_back(HUGE_VAL);
            M.bias_S.assign(m->bias_strengths, m->bias_strengths + m->n_bias_heights + 1);
            M.app_delays.assign(m->application_delays, m->application_delays + M.E);
            M.delay_type = m->delay_type;
        }
        if (m->n_rate_segments > 0) {
            /* RecombinationBias::set_model_rates (pfparam.hpp:199-211): segments that start inside the locus */
            M.guided = true;
            for (int k = 0; k < m->n_rate_segments; ++k) {
                if (!(m->rate_positions[k] < M.L)) break;
                M.seg_pos.push_back(m->rate_positions[k]);
                M.seg_rho.push_back(m->rate_values[k]);
                for (int i = 0; i < M.n; ++i) M.leaf_rate.push_back(m->leaf_rel_rates[(size_t)k * M.n + i]);
            }
            if (M.seg_pos.empty() || M.seg_pos[0] != 0.0) throw std::runtime_error("recombination guide must start at position 0");
            if (!M.biased) {
                if (!m->application_delays) throw std::runtime_error("a recombination guide needs application_delays");
                M.bias_H = {0.0, HUGE_VAL};
                M.bias_S = {1.0};
                M.app_delays.assign(m->application_delays, m->application_delays + M.E);
                M.delay_type = m->delay_type;
            }
        }
        f->Np = p->np; f->ess_fraction = p->ess_fraction; f->seed = p->seed;
        f->max_trace_events = p->max_trace_events;
        return f;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void smco_destroy(void* h) { delete (Filter*)h; }

int smco_load_lookahead(void* h, const smco_lookahead* la) {
    GUARD(
        Filter* f = (Filter*)h;
        const int n = f->M.n;
        const int64_t S = la->n;
        if (la->level < 0 || la->level > 4) throw std::runtime_error("-apf must be 0..4");
        f->apf = la->level; f->la_D = la->max_doubletons; f->la_Q = la->n_quantiles;
        f->la_fsd.assign(la->first_singleton_distance, la->first_singleton_distance + S * n);
        f->la_rmr.assign(la->relative_mutation_rate, la->relative_mutation_rate + S * n);
        f->la_unph.assign(la->is_singleton_unphased, la->is_singleton_unphased + S * n);
        f->la_nd.assign(la->n_doubletons, la->n_doubletons + S);
        f->la_didx.assign(la->doubleton_idx, la->doubleton_idx + S * f->la_D * 4);
        f->la_ddist.assign(la->doubleton_dist, la->doubleton_dist + S * f->la_D * 2);
        f->la_split.assign(la->first_split_distance, la->first_split_distance + S);
        f->la_salleles.assign(la->split_alleles, la->split_alleles + S * n);
        f->la_sk.assign(la->split_count, la->split_count + S);
        f->la_q.assign(la->quantiles, la->quantiles + f->la_Q);
        f->la_tbl.assign(la->tbl_lengths, la->tbl_lengths + (int64_t)n * f->la_Q);
        f->la_mean_tbl = la->mean_total_branch_length;
    )
}

int smco_terminal_branch_quantiles(const smco_model* m, uint64_t seed, int64_t n_trees, const double* quantiles, int32_t nq,
                                   double* lengths_out, double* mean_total_out) {
    GUARD(
        Filter f;
        fill_model(f.M, m);
        f.M.recflags.assign(f.M.E, 3);
        f.M.lags.assign(f.M.E, 0.0);
        const int n = f.M.n;
        f.Np = 1; f.seed = seed; f.stream = 3; f.record_events = false;
        f.rng.assign(1, SlotRng{0, 0.0});
        std::vector<std::vector<double>> tbls(n);
        for (auto& v : tbls) v.reserve(n_trees);
        std::vector<double> lengths(n_trees);
        for (int64_t r = 0; r < n_trees; ++r) {
            Particle p;
            p.head.assign(f.M.E, nullptr);
            f.rng[0] = SlotRng{0, 0.0};
            f.slot_override = r;
            f.rng[0].ebuf = -smc_log(f.uni(0));
            f.build_initial_tree(0, p);
            lengths[r] = p.Ltree;
            for (int i = 0; i < n; ++i) {
                int pr = -1;
                for (int k = 0; k < n - 1 && pr < 0; ++k)
                    if (p.tr.C[k][0] == i || p.tr.C[k][1] == i) pr = k;
                tbls[i].push_back(p.tr.S[pr]);          /* parent_height_ignoring_migrations (smcsmc.cpp:115-125) */
            }
        }
        for (int i = 0; i < n; ++i) {
            std::sort(tbls[i].begin(), tbls[i].end());
            for (int q = 0; q < nq; ++q) lengths_out[i * nq + q] = tbls[i][(int64_t)(quantiles[q] * (double)n_trees)];
        }
        double total_length = 0.0;                      /* serial, in tree order (smcsmc.cpp:145) */
        for (int64_t r = 0; r < n_trees; ++r) total_length += lengths[r];
        *mean_total_out = total_length / (double)n_trees;
    )
}

int smco_init_prior(void* h, double initial_position) { GUARD(((Filter*)h)->init_prior(initial_position);) }

int smco_update_segment(void* h, const smco_segments* segs, int64_t s) {
    GUARD(Filter* f = (Filter*)h; f->update_segment(segs, s); f->tr_T.push_back(f->last_T); f->tr_logl.push_back(f->logl);)
}
int smco_count(void* h, double current_base, int end_data) { GUARD(((Filter*)h)->count(current_base, end_data != 0);) }
int smco_resample(void* h, double update_to) {
    try { Filter* f = (Filter*)h; int r = f->resample(update_to); f->seg_done++; return r; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int smco_finish(void* h) { GUARD(((Filter*)h)->finish();) }

int smco_run(void* h, const smco_segments* sg) {
    GUARD(
        Filter* f = (Filter*)h;
        for (int64_t s = 0; s < sg->n; ++s) {
            f->update_segment(sg, s);
            f->tr_T.push_back(f->last_T);
            f->tr_logl.push_back(f->logl);
            double pos = std::min(sg->start[s] + sg->length[s], f->M.L);
            f->count(pos, false);
            f->resample(pos);
            f->seg_done++;
            if (sg->start[s] + sg->length[s] >= f->M.L) break;
        }
        f->finish();
    )
}

int64_t smco_num_segments_done(void* h) { return ((Filter*)h)->seg_done; }

int smco_get_trace(void* h, double* T, double* ess, int32_t* resampled, double* logl, int64_t n) {
    Filter* f = (Filter*)h;
    int64_t m = std::min<int64_t>(n, (int64_t)f->tr_T.size());
    for (int64_t i = 0; i < m; ++i) {
        if (T) T[i] = f->tr_T[i];
        if (logl) logl[i] = f->tr_logl[i];
        if (ess) ess[i] = i < (int64_t)f->tr_ess.size() ? f->tr_ess[i] : 0;
        if (resampled) resampled[i] = i < (int64_t)f->tr_flag.size() ? f->tr_flag[i] : 0;
    }
    return (int)m;
}

int smco_get_resample_events(void* h, int32_t* seg_idx, int32_t* parents, int32_t max_events) {
    Filter* f = (Filter*)h;
    int n = std::min<int>(max_events, (int)f->ev_parents.size());
    for (int k = 0; k < n; ++k) {
        if (seg_idx) seg_idx[k] = f->ev_seg[k];
        if (parents) std::copy(f->ev_parents[k].begin(), f->ev_parents[k].end(), parents + (int64_t)k * f->Np);
    }
    return n;
}

int smco_get_particles(void* h, double* w_post, double* w_pilot, double* heights, int8_t* children, double* next_base) {
    Filter* f = (Filter*)h;
    const int n = f->M.n;
    for (int64_t i = 0; i < f->Np; ++i) {
        const Particle& p = f->parts[i];
        if (w_post) w_post[i] = p.w_post;
        if (w_pilot) w_pilot[i] = p.w_pilot;
        if (next_base) next_base[i] = p.next_base;
        for (int r = 0; r < n - 1; ++r) {
            if (heights) heights[i * (n - 1) + r] = p.tr.S[r];
            if (children) { children[(i * (n - 1) + r) * 2] = p.tr.C[r][0]; children[(i * (n - 1) + r) * 2 + 1] = p.tr.C[r][1]; }
        }
    }
    return 0;
}

int smco_get_counts(void* h, double* out, int32_t n) {
    Filter* f = (Filter*)h;
    const int E = f->M.E, P = f->M.P;
    if (n < SMCO_COUNTS_LEN2(E, P)) return -1;
    double* o = out;
    for (int i = 0; i < E * P; ++i) o[i] = f->coal_count[i];
    o += E * P;
    for (int i = 0; i < E * P; ++i) o[i] = f->coal_opp[i];
    o += E * P;
    for (int i = 0; i < E * P; ++i) o[i] = f->coal_w2[i];
    o += E * P;
    for (int e = 0; e < E; ++e) { o[e] = f->rec_count[e]; o[E + e] = f->rec_opp[e]; o[2 * E + e] = f->rec_w2[e]; }
    o += 3 * E;
    if (P > 1) {
        for (int i = 0; i < E * P * P; ++i) o[i] = f->mig_count[i];
        o += E * P * P;
        for (int i = 0; i < E * P; ++i) o[i] = f->mig_opp[i];
        o += E * P;
        for (int i = 0; i < E * P; ++i) o[i] = f->mig_w2[i];
        o += E * P;
    }
    o[0] = f->delayed_opp; o[1] = f->delayed_count;
    o[2] = (double)f->n_resample; o[3] = f->logl;
    return 0;
}

int smco_enable_local_recomb(void* h) { ((Filter*)h)->local_map = true; return 0; }

/* -arg: call before smco_init_prior */
int smco_enable_tree_recording(void* h) {
    Filter* f = (Filter*)h;
    f->record_trees = true;
    return 0;
}

/* ParticleContainer::printTrees (pc.cpp:515-555) after resample(..., NULL, 1) (smcsmc.cpp:395): the particle whose
 * cumulative posterior weight passes U * total, U the next uniform of the resampler's stream; its events last first */
int64_t smco_sample_tree_events(void* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int64_t max_events,
                                int64_t* particle_out) {
    return smco_sample_tree_events_pops(h, kind, pos, height, desc, nullptr, nullptr, max_events, particle_out);
}

int64_t smco_sample_tree_events_pops(void* h, int32_t* kind, double* pos, double* height, uint32_t* desc, int32_t* from_pop,
                                     int32_t* to_pop, int64_t max_events, int64_t* particle_out) {
    Filter* f = (Filter*)h;
    if (!f->record_trees) return -1;
    double total = 0.0;
    for (int64_t i = 0; i < f->Np; ++i) total += f->parts[i].w_pilot;   /* pilotWeight(), pc.cpp:256-262 */
    const double u = philox_uniform(f->seed, 0xFFFFFFFFu, 1, (uint64_t)f->n_resample);
    int64_t j = 0;
    double acc = 0.0;
    for (; j < f->Np - 1; ++j) { acc += f->parts[j].w_pilot; if (acc > u * total) break; }
    if (particle_out) *particle_out = j;
    int64_t n = 0;
    for (const TreeEv* ev = f->parts[j].tree_head.get(); ev; ev = ev->parent.get()) {
        if (n < max_events) {
            if (kind) kind[n] = ev->kind;
            if (pos) pos[n] = ev->x;
            if (height) height[n] = ev->t;
            if (desc) desc[n] = ev->desc;
            if (from_pop) from_pop[n] = ev->from_pop;
            if (to_pop) to_pop[n] = ev->to_pop;
        }
        ++n;
    }
    return n;
}

/* opp_diff[nbins], counts[(nsam+2)*nbins] */
int smco_get_local_recomb(void* h, double* opp_diff, double* counts, int64_t nbins) {
    Filter* f = (Filter*)h;
    if (!f->local_map) return -1;
    for (int64_t b = 0; b < nbins; ++b) {
        opp_diff[b] = b < (int64_t)f->local_opp.size() ? f->local_opp[b] : 0.0;
        for (int k = 0; k < f->M.n + 2; ++k)
            counts[(int64_t)k * nbins + b] = b < (int64_t)f->local_cnt[k].size() ? f->local_cnt[k][b] : 0.0;
    }
    return 0;
}

int smco_get_migrations(void* h, int32_t* nm, double* times, int8_t* branch, int8_t* newpop, int8_t* node_pops, int32_t cap) {
    Filter* f = (Filter*)h;
    const int n = f->M.n;
    for (int64_t i = 0; i < f->Np; ++i) {
        const Tree& t = f->parts[i].tr;
        if (nm) nm[i] = t.nm;
        for (int m = 0; m < cap; ++m) {
            if (times) times[i * cap + m] = m < t.nm ? t.Mt[m] : 0.0;
            if (branch) branch[i * cap + m] = m < t.nm ? t.Mb[m] : 0;
            if (newpop) newpop[i * cap + m] = m < t.nm ? t.Mq[m] : 0;
        }
        if (node_pops) for (int r = 0; r < n - 1; ++r) node_pops[i * (n - 1) + r] = f->M.P > 1 ? t.Pn[r] : 0;
    }
    return 0;
}

double smco_logl(void* h) { return ((Filter*)h)->logl; }

int smco_get_stats(void* h, int64_t* n_recomb, int64_t* n_alloc, int64_t* n_resamples) {
    Filter* f = (Filter*)h;
    if (n_recomb) *n_recomb = f->n_recomb;
    if (n_alloc) *n_alloc = f->pool.n_alloc;
    if (n_resamples) *n_resamples = f->n_resample;
    return 0;
}

int smco_get_delay_stats(void* h, int64_t* n_forced, int32_t* peak_pending) {
    Filter* f = (Filter*)h;
    if (n_forced) *n_forced = f->n_delay_evict;
    if (peak_pending) *peak_pending = f->delay_peak;
    return 0;
}

int smco_median_survival(const smco_model* m, uint64_t seed, int32_t min_events, int64_t max_trees, double* median_out,
                         int64_t* trees_used) {
    GUARD(
        Model M;
        fill_model(M, m);
        M.recflags.assign(M.E, 3);
        M.lags.assign(M.E, 0.0);
        median_survival(M, seed, min_events, max_trees, median_out, trees_used);
    )
}

double smco_exp(double x) { return smc_exp(x); }
double smco_log(double x) { return smc_log(x); }
double smco_fastexp(double x) { return fastexp(x); }
double smco_uniform(uint64_t seed, uint32_t slot, uint32_t stream, uint64_t draw) { return philox_uniform(seed, slot, stream, draw); }
double smco_canon_sum(const double* x, int64_t n) { return canon_sum(x, n); }
void smco_canon_scan(const double* x, double* incl, int64_t n) { canon_scan(x, incl, n); }
void smco_systematic(const double* pilot, int64_t n, double u, int32_t* lo) {
    std::vector<double> incl(n);
    canon_scan(pilot, incl.data(), n);
    systematic(incl.data(), n, u, lo);
}

}  // extern "C"
