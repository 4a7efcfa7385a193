// smcsmc_amd/csrc/host/main.cpp -- the drop-in `smcsmc` binary: main + pfARG_core
// (/root/reference/src/smcsmc.cpp:46-103, 278-401) on top of the HIP C-ABI (include/smcsmc_pf.h).
#include <zlib.h>

#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "../../../include/smcsmc_pf.h"
#include "smcsmc_host.hpp"
#include "mgpu.hpp"

using namespace std;

static void dump_model_json(const PfParam& p) {
    const HostModel& m = p.model;
    cout << setprecision(17);
    cout << "{\"N0\": " << m.N0 << ", \"nsam\": " << m.nsam << ", \"npop\": " << m.npop << ", \"loci_length\": " << m.loci_length
         << ", \"mutation_rate\": " << m.mutation_rate << ", \"recombination_rate\": " << m.recombination_rate
         << ", \"vb\": " << (m.vb ? "true" : "false") << ", \"seed\": " << m.seed << ", \"Np\": " << p.particles
         << ", \"change_times\": [";
    for (size_t e = 0; e < m.change_times.size(); ++e) cout << (e ? ", " : "") << m.change_times[e];
    cout << "], \"pop_sizes\": [";
    for (size_t e = 0; e < m.pop_sizes.size(); ++e) {
        cout << (e ? ", [" : "[");
        for (size_t k = 0; k < m.pop_sizes[e].size(); ++k) cout << (k ? ", " : "") << m.pop_sizes[e][k];
        cout << "]";
    }
    cout << "], \"mig_rates\": [";
    for (size_t e = 0; e < m.mig_rates.size(); ++e) {
        cout << (e ? ", [" : "[");
        for (size_t k = 0; k < m.mig_rates[e].size(); ++k) cout << (k ? ", " : "") << m.mig_rates[e][k];
        cout << "]";
    }
    cout << "], \"single_mig\": [";
    for (size_t e = 0; e < m.single_mig.size(); ++e) {
        cout << (e ? ", [" : "[");
        for (size_t k = 0; k < m.single_mig[e].size(); ++k) cout << (k ? ", " : "") << m.single_mig[e][k];
        cout << "]";
    }
    cout << "], \"sample_pops\": [";
    for (size_t k = 0; k < m.sample_pops.size(); ++k) cout << (k ? ", " : "") << m.sample_pops[k];
    cout << "], \"record_mask\": [";
    for (size_t k = 0; k < p.record_mask.size(); ++k) cout << (k ? ", " : "") << p.record_mask[k];
    cout << "], \"guide_positions\": [";
    for (size_t k = 0; k < p.guide_positions.size(); ++k) cout << (k ? ", " : "") << p.guide_positions[k];
    cout << "], \"guide_rates\": [";
    for (size_t k = 0; k < p.guide_rates.size(); ++k) cout << (k ? ", " : "") << p.guide_rates[k];
    cout << "], \"guide_leaf_rates\": [";
    for (size_t k = 0; k < p.guide_leaf_rates.size(); ++k) cout << (k ? ", " : "") << p.guide_leaf_rates[k];
    cout << "]}" << endl;
}

static void dump_lookahead_json(const PfParam& p) {
    LookaheadArrays la;
    p.segments->pack_lookahead(la);
    const size_t S = la.n_doubletons.size();
    const int n = p.model.nsam, D = la.max_doubletons;
    cout << setprecision(17) << "{\"max_doubletons\": " << D << ", \"rows\": [";
    for (size_t i = 0; i < S; ++i) {
        cout << (i ? ", " : "") << "{\"fsd\": [";
        for (int j = 0; j < n; ++j) cout << (j ? ", " : "") << la.first_singleton_distance[i * n + j];
        cout << "], \"rmr\": [";
        for (int j = 0; j < n; ++j) cout << (j ? ", " : "") << la.relative_mutation_rate[i * n + j];
        cout << "], \"unph\": [";
        for (int j = 0; j < n; ++j) cout << (j ? ", " : "") << (int)la.is_singleton_unphased[i * n + j];
        cout << "], \"dbl\": [";
        for (int k = 0; k < la.n_doubletons[i]; ++k) {
            const int8_t* di = &la.doubleton_idx[(i * D + k) * 4];
            cout << (k ? ", " : "") << "[" << (int)di[0] << ", " << (int)di[1] << ", " << (int)di[2] << ", " << (int)di[3] << ", "
                 << la.doubleton_dist[(i * D + k) * 2] << ", " << la.doubleton_dist[(i * D + k) * 2 + 1] << "]";
        }
        cout << "], \"split\": " << la.first_split_distance[i] << ", \"split_alleles\": [";
        for (int j = 0; j < n; ++j) cout << (j ? ", " : "") << (int)la.split_alleles[i * n + j];
        cout << "], \"split_count\": " << la.split_count[i] << "}";
    }
    cout << "]}" << endl;
}

// -dumpsegments: the arrays pf_load_segments would receive, with the constant -lag or the reference's uncalibrated
// per-epoch default 4 / (rho * top of epoch) (count.cpp:230-247) as lags -- no device needed
static void dump_segments_json(const PfParam& p) {
    const HostModel& M = p.model;
    const size_t E = M.change_times.size();
    std::vector<double> lags(E, 20000.0);
    for (size_t e = 0; e < E && E > 1; ++e)
        lags[e] = p.lag > 0 ? p.lag : 4.0 / (M.recombination_rate * M.change_times[std::min(e + 1, E - 1)]);
    std::vector<double> start, length;
    std::vector<int8_t> state, alleles;
    std::vector<int32_t> limit;
    p.segments->pack(lags, start, length, state, alleles, limit);
    cout << setprecision(17) << "{\"nsam\": " << M.nsam << ", \"lags\": [";
    for (size_t e = 0; e < E; ++e) cout << (e ? ", " : "") << lags[e];
    auto list = [&](const char* name, auto&& v) {
        cout << "], \"" << name << "\": [";
        for (size_t k = 0; k < v.size(); ++k) cout << (k ? ", " : "") << (double)v[k];
    };
    list("start", start); list("length", length); list("state", state); list("alleles", alleles); list("max_record_epoch", limit);
    cout << "]}" << endl;
}

static void pf_check(int rc) {
    if (rc < 0) throw std::runtime_error(pf_last_error());
}

// The reference carries one generator across its EM iterations (pfparam.cpp:316); with counter-based streams every
// iteration gets its own key instead, derived from one base seed that is resolved once per process.
static uint64_t base_seed(const HostModel& M) {
    static const uint64_t from_clock = (uint64_t)time(nullptr);
    return M.seed_set ? M.seed : from_clock;
}

// what the chunks of one E-step share: the lag calibration depends on the model only (done once per iteration), and
// every chunk hands its raw sufficient statistics back instead of writing the rows itself
struct ChunkJob {
    const std::vector<double>* survival = nullptr;     // median survival distances per epoch, or null: calibrate here
    std::vector<double>* survival_out = nullptr;       // receives them when they were calibrated here
    std::vector<double>* packed_out = nullptr;         // PF_COUNTS_LEN2 doubles, raw sums (no pseudo-counts)
    uint64_t seed_offset = 0;                          // + chunk index, the rule of smcsmc_amd/em.py
    int count_wgs = 0;                                 // pf_params.count_wgs (0: the library's default)
};

// One chunk's filter between its creation and the collection of its results: pfARG_core (smcsmc.cpp:278-401) in three
// steps -- open_filter (everything up to the initial particles), the rows (pf_run, or pf_run_many for several chunks in
// lockstep on one device), close_filter (final flush, outputs, statistics).
struct ChunkFilter {
    PfParam* P = nullptr;
    pf_handle* h = nullptr;
    int E = 0, NP = 1;
    int64_t rows = 0;
    std::vector<double> start, length;       // row table as loaded (progress report, .resample)
    ~ChunkFilter() { if (h) pf_destroy(h); }
};

static void release_filter(ChunkFilter& F) {
    if (F.h) { pf_destroy(F.h); F.h = nullptr; }
}

static void open_filter(ChunkFilter& F, PfParam& P, const HostModel& M0, int device, const ChunkJob* job) {
    F.P = &P;
    HostModel& M = P.model;
    const int E = (int)M.change_times.size();
    const int NP = M.npop;
    if (NP > 4) throw Unsupported("models with more than four populations");

    std::vector<double> pop_sizes((size_t)E * NP), mig_rates((size_t)E * NP * NP), single_mig((size_t)E * NP * NP);
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < NP; ++a) {
            pop_sizes[(size_t)e * NP + a] = M.pop_sizes[e][a];
            for (int b = 0; b < NP; ++b) {
                mig_rates[((size_t)e * NP + a) * NP + b] = M.mig_rates[e][(size_t)a * NP + b];
                single_mig[((size_t)e * NP + a) * NP + b] = M.single_mig[e][(size_t)a * NP + b];
            }
        }
    std::vector<int32_t> sample_pops(M.sample_pops.begin(), M.sample_pops.end());
    pf_model pm;
    memset(&pm, 0, sizeof(pm));
    pm.n_epochs = E; pm.n_pops = NP; pm.nsam = M.nsam;
    if (NP > 1) { pm.mig_rates = mig_rates.data(); pm.single_mig = single_mig.data(); pm.sample_pops = sample_pops.data(); }
    std::vector<double> vb_cc, vb_mc;
    if (M.vb) {          // variational_bayes_correction_ (particle.cpp:266-272)
        for (int e = 0; e < E; ++e) {
            for (int a = 0; a < NP; ++a) vb_cc.push_back(M.coal_counts[e][a]);
            for (int k = 0; k < NP * NP; ++k) vb_mc.push_back(M.mig_counts[e][k]);
        }
        pm.vb_coal_counts = vb_cc.data(); pm.vb_mig_counts = vb_mc.data();
    }
    pm.flags = (P.ancestral_aware ? 1 : 0) | (P.dephase ? 2 : 0);
    pm.loci_length = M.loci_length; pm.mutation_rate = M.mutation_rate; pm.recombination_rate = M.recombination_rate;
    pm.change_times = M.change_times.data(); pm.pop_sizes = pop_sizes.data();
    pm.record_flags = P.record_mask.data();

    // lags: CountModel::init_lags (count.cpp:230-247) then reset_lag with the calibrated survival (261-265)
    std::vector<double> lags(E);
    if (E == 1) lags[0] = 20000;
    else
        for (int e = 0; e < E; ++e) {
            double tmax = e == E - 1 ? M.change_times[E - 1] : M.change_times[e + 1];
            lags[e] = P.lag > 0 ? P.lag : 4.0 / (M0.recombination_rate * tmax);   // CountModel keeps its initial model
        }
    // calculate_median_survival_distances is always run by the reference (smcsmc.cpp:287); its result feeds the
    // lags (when calibrating) and the application delays of the importance weights (smcsmc.cpp:306-307)
    const bool biased = !M.bias_heights.empty();
    const bool guided = !P.guide_positions.empty();          // the calibration runs at the true rate (smcsmc.cpp:287-291)
    std::vector<double> med(E, 0.0), app_delays(E, 0.0);
    if (job && job->survival && !job->survival->empty()) med = *job->survival;
    else if (P.calibrate_lag || biased || guided) {
        int64_t trees = 0;
        pm.lags = lags.data();
        pf_check(pf_median_survival(&pm, 1, 200, 1000000, med.data(), &trees, device));
        for (int e = 0; e < E; ++e) clog << " Epoch " << e << ": survival distance " << med[e] << endl;
    }
    if (job && job->survival_out) *job->survival_out = med;
    if (P.calibrate_lag)
        for (int e = 0; e < E; ++e) lags[e] = med[e] * P.lag_fraction;      // reset_lag, count.cpp:261-265
    if (biased || guided) {
        for (int e = 0; e < E; ++e) {
            app_delays[e] = med[e] * P.delay;                                 // Model::lags_to_application_delays
            if (P.delay > 0) clog << " Application delay for epoch " << e << " set to " << app_delays[e] << endl;
        }
        pm.delay_type = P.delay_type | (P.delay_all ? 4 : 0);
        pm.application_delays = app_delays.data();
    }
    if (guided) {          // PfParam::setModelRates (pfparam.cpp:383-388)
        cout << "Setting model rates" << endl;
        pm.n_rate_segments = (int32_t)P.guide_positions.size();
        pm.rate_positions = P.guide_positions.data();
        pm.rate_values = P.guide_rates.data();
        pm.leaf_rel_rates = P.guide_leaf_rates.data();
    }
    if (biased) {
        pm.n_bias_heights = (int32_t)M.bias_heights.size();
        pm.delay_type = P.delay_type | (P.delay_all ? 4 : 0);
        pm.bias_heights = M.bias_heights.data();
        pm.bias_strengths = M.bias_strengths.data();
        pm.application_delays = app_delays.data();
    }
    pm.lags = lags.data();
    clog << "    Lags set to:";
    for (double l : lags) clog << " " << l;
    clog << endl;
    clog << " Starting position: " << fixed << setprecision(0) << P.start_position << setprecision(6) << scientific << endl;

    std::vector<double> start, length;
    std::vector<int8_t> state, alleles;
    std::vector<int32_t> mre;
    P.segments->pack(lags, start, length, state, alleles, mre);
    if (P.record_all) std::fill(mre.begin(), mre.end(), E - 1);
    pf_segments sg = {(int64_t)start.size(), start.data(), length.data(), state.data(), alleles.data(), mre.data()};

    // auxiliary particle filter: look-ahead per row + terminal branch length quantiles (smcsmc.cpp:288, 128-166)
    LookaheadArrays la;
    std::vector<double> tbl_lengths;
    double mean_tbl = 0;
    const double tbl_quantiles[7] = {0.001, 0.003, 0.01, 0.03, 0.1, 0.5, 0.95};
    if (P.apf_level > 0) {
        if (P.segments->empty_file()) throw Unsupported("-apf without -seg data");
        cout << "Calculating terminal branch length quantiles..." << endl;
        tbl_lengths.resize((size_t)M.nsam * 7);
        pf_check(pf_terminal_branch_quantiles(&pm, 1, 1000000, tbl_quantiles, 7, tbl_lengths.data(), &mean_tbl, device));
        for (int i = 0; i < M.nsam; ++i) {
            cout << "Lineage " << i << "; Terminal branch length quantiles:";
            for (int q = 0; q < 7; ++q) cout << " [" << tbl_quantiles[q] << ":] " << tbl_lengths[(size_t)i * 7 + q];
            cout << endl;
        }
        P.segments->pack_lookahead(la);
    }

    pf_params pp;
    memset(&pp, 0, sizeof(pp));
    pp.np = (int64_t)P.particles; pp.ess_fraction = P.ess_fraction;
    pp.seed = base_seed(M) + 1000ull * (uint64_t)P.em_iteration + (job ? job->seed_offset : 0);   // same rule as smcsmc_amd/em.py: seed + 1000 * iteration + chunk
    pp.max_trace_events = 0;
    pp.flags = 1;          // the local recombination map is always recorded (smcsmc.cpp:376-383)
    pp.mig_cap = P.mig_cap;
    pp.delay_cap = P.delay_cap;
    pp.log_cap = P.log_cap;
    pp.count_wgs = job ? job->count_wgs : P.count_wgs;
    if (P.delay_evict) pp.flags |= 4;
    if (P.record_trees) {
        if (NP > 1 && M.nsam > 8) throw Unsupported("-arg with more than one population and more than 8 samples");
        pp.flags |= 2;     // -arg (pfparam.cpp:353-357)
        // nothing may be overwritten while the history is needed: a slot appends about 0.6 records per row for its
        // recombinations and up to one per resampling, and there are at most as many generations as rows
        auto pow2_at_least = [](double v) { long long c = 16384; while ((double)c < v) c <<= 1; return c; };
        // (recombinations per row grow with the tree length: the figure above is for four samples)
        const long long log_cap = pow2_at_least(1.4 * (double)start.size() * std::max(1.0, (M.nsam - 1) / 3.0));
        const long long gen_cap = pow2_at_least((double)start.size() + 2.0);
        const double gib = ((double)P.particles * (double)log_cap * (5.0 + M.nsam - 1) * 8.0 + (double)gen_cap * (double)P.particles * 20.0) / (1024.0 * 1024.0 * 1024.0);
        clog << " -arg: event log of " << log_cap << " records per particle, " << gen_cap << " generations (" << fixed << setprecision(1)
             << gib << " GiB)" << setprecision(6) << scientific << endl;
        pp.log_cap = log_cap;
        pp.gen_cap = gen_cap;
    }
    pf_handle* h = pf_create(&pm, &pp, device);
    if (!h) throw std::runtime_error(pf_last_error());
    F.h = h; F.E = E; F.NP = NP;
    pf_check(pf_load_segments(h, &sg));
    if (P.apf_level > 0) {
        pf_lookahead pl;
        memset(&pl, 0, sizeof(pl));
        pl.level = P.apf_level; pl.max_doubletons = la.max_doubletons; pl.n_quantiles = 7; pl.n = sg.n;
        pl.first_singleton_distance = la.first_singleton_distance.data();
        pl.relative_mutation_rate = la.relative_mutation_rate.data();
        pl.is_singleton_unphased = la.is_singleton_unphased.data();
        pl.n_doubletons = la.n_doubletons.data();
        pl.doubleton_idx = la.doubleton_idx.data(); pl.doubleton_dist = la.doubleton_dist.data();
        pl.first_split_distance = la.first_split_distance.data();
        pl.split_alleles = la.split_alleles.data(); pl.split_count = la.split_count.data();
        pl.quantiles = tbl_quantiles; pl.tbl_lengths = tbl_lengths.data(); pl.mean_total_branch_length = mean_tbl;
        pf_check(pf_load_lookahead(h, &pl));
    }
    pf_check(pf_init_prior(h, start.empty() ? 0.0 : start[0]));
    F.rows = sg.n;
    F.start.swap(start); F.length.swap(length);
}

// the do-while of pfARG_core (smcsmc.cpp:324-360) for the filters of `group` (one, or several chunks of one device in
// lockstep: one launch per row covers all of them, pf_run_many)
static void run_filters(std::vector<ChunkFilter*>& group, bool progress) {
    int64_t S = 0;
    for (ChunkFilter* F : group) S = std::max(S, F->rows);
    std::vector<pf_handle*> hs;
    for (ChunkFilter* F : group) hs.push_back(F->h);
    const ChunkFilter& F0 = *group[0];
    const double L0 = F0.P->model.loci_length;
    const int64_t step = 1000;
    for (int64_t s0 = 0; s0 < S; s0 += step) {
        const int64_t s1 = std::min(S, s0 + step);
        if (hs.size() == 1) pf_check(pf_run(hs[0], s0, std::min(s1, F0.rows)));
        else pf_check(pf_run_many(hs.data(), (int32_t)hs.size(), s0, s1));
        if (progress) {
            const int64_t r = std::min(s1, F0.rows);
            const double end = r > 0 ? F0.start[r - 1] + F0.length[r - 1] : 0.0;
            cout << "\r Particle filtering " << setw(4) << int((end * 100) / L0) << "% completed." << flush;
            if (hs.size() == 1 && end >= L0) break;
        }
    }
}

static void report_counts(PfParam& P, const HostModel& M0, const std::vector<double>& packed, double chunks);

static void close_filter(ChunkFilter& F, const HostModel& M0, const ChunkJob* job) {
    PfParam& P = *F.P;
    HostModel& M = P.model;
    pf_handle* h = F.h;
    const int E = F.E, NP = F.NP;
    const std::vector<double>& start = F.start;
    const std::vector<double>& length = F.length;
    {
        pf_check(pf_finish(h));
        cout << "\r Particle filtering step 100% completed." << endl;
        int64_t done = pf_num_segments_done(h);
        std::vector<double> packed(PF_COUNTS_LEN2(E, NP));
        pf_check(pf_get_counts(h, packed.data(), (int32_t)packed.size()));
        const double* tail = &packed[packed.size() - 4];
        clog << "Got to end of sequence; resampled " << (long long)tail[2] << " times" << endl;
        {
            // the reference's heap of delayed factors is unbounded (particle.hpp:248); the store here is not, so say how full it got
            int64_t forced = 0; int32_t peak = 0;
            pf_check(pf_get_delay_stats(h, &forced, &peak));
            if (peak > 0)
                clog << " Delayed importance factors: at most " << peak << " pending per particle; " << (long long)forced
                     << " applied early to make room" << endl;
        }
        clog << " Inference step completed." << endl;
        // a chunk of a multi-chunk E-step: the statistics go to the reduction; its local recombination map is written like any
        // chunk process's (<prefix>.chunkN.recomb.gz: the reference writes one per chunk process, smcsmc.cpp:376-383)
        const bool chunk_job = job && job->packed_out;
        if (chunk_job) *job->packed_out = packed;
        if (!chunk_job && P.write_resample) {
            std::vector<double> ess(done);
            std::vector<int32_t> flag(done);
            pf_check(pf_get_trace(h, nullptr, ess.data(), flag.data(), nullptr, done));
            for (int64_t s = 0; s < done; ++s)
                if (flag[s]) P.write_resample_row(std::min(start[s] + length[s], M.loci_length), ess[s]);
        }
        // <prefix>.recomb.gz (smcsmc.cpp:376-383): CountModel::dump_local_recomb_logs (count.cpp:616-654)
        {
            const int64_t nb = (int64_t)(M.loci_length / 100.0);
            std::vector<double> lopp(nb), lcnt((size_t)(M.nsam + 2) * nb);
            pf_check(pf_get_local_recomb(h, lopp.data(), lcnt.data(), nb));
            // One million rows of eight numbers for a 100 Mb chunk: formatted and deflated in slices by a few threads,
            // each slice a complete gzip member; members written in order form a valid .gz (zcat, Python's gzip and
            // the boost reader of the front-end all read multi-member files).
            std::vector<double> cumopp(nb);
            {
                double cur = 0.0;
                for (int64_t idx = 0; idx < nb; ++idx) { cur += lopp[idx]; cumopp[idx] = cur; }
            }
            const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)std::thread::hardware_concurrency(), 16), nb / 4096 + 1));
            std::vector<std::string> member(nthr);
            std::vector<int> zrc(nthr, 0);
            auto work = [&](int t) {
                const int64_t i0 = nb * t / nthr, i1 = nb * (t + 1) / nthr;
                std::string text;
                text.reserve((size_t)(i1 - i0) * (24 + 12 * (M.nsam + 3)) + 256);
                char buf[64];
                if (t == 0 && P.em_iteration == 0) {
                    text += "iter\tlocus\tsize\topp_per_nt";
                    for (int sdx = 0; sdx < M.nsam; ++sdx) { snprintf(buf, sizeof buf, "\t%d", sdx + 1); text += buf; }
                    text += "\ttime\tlog_time\n";
                }
                for (int64_t idx = i0; idx < i1; ++idx) {
                    // same characters as operator<< with fixed/setprecision(0) and scientific/setprecision(5)
                    int len = snprintf(buf, sizeof buf, "%zu\t%.0f\t%.0f\t%.5e", P.em_iteration, idx * 100.0 + P.start_position, 100.0,
                                       cumopp[idx] / 100.0);
                    text.append(buf, (size_t)len);
                    for (int k = 0; k < M.nsam + 2; ++k) {
                        len = snprintf(buf, sizeof buf, "\t%.5e", lcnt[(size_t)k * nb + idx] / 100.0);
                        text.append(buf, (size_t)len);
                    }
                    text += '\n';
                }
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { zrc[t] = 1; return; }
                std::string& out = member[t];
                out.resize(deflateBound(&zs, (uLong)text.size()) + 64);
                zs.next_in = (Bytef*)text.data(); zs.avail_in = (uInt)text.size();
                zs.next_out = (Bytef*)&out[0]; zs.avail_out = (uInt)out.size();
                if (deflate(&zs, Z_FINISH) != Z_STREAM_END) zrc[t] = 1;
                out.resize(zs.total_out);
                deflateEnd(&zs);
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < nthr; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto& th : pool) th.join();
            bool ok = true;
            for (int t = 0; t < nthr; ++t) ok = ok && zrc[t] == 0;
            FILE* fz = ok ? fopen(P.recomb_map_path.c_str(), "ab") : nullptr;
            if (fz) {
                for (int t = 0; t < nthr; ++t) fwrite(member[t].data(), 1, member[t].size(), fz);
                fclose(fz);
            }
        }
        if (chunk_job) return;
        // <prefix>.trees.gz (ParticleContainer::printTrees, pc.cpp:515-555) after the one-particle draw of smcsmc.cpp:395
        if (P.record_trees) {          // every E-step overwrites it, as the reference does
            int64_t particle = 0;
            const int64_t nev = pf_sample_tree_events_pops(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &particle);
            if (nev < 0) pf_check(-1);
            std::vector<int32_t> kind((size_t)nev), from((size_t)nev), to((size_t)nev);
            std::vector<double> xs((size_t)nev), ts((size_t)nev);
            std::vector<uint32_t> ds((size_t)nev);
            if (pf_sample_tree_events_pops(h, kind.data(), xs.data(), ts.data(), ds.data(), from.data(), to.data(), nev, &particle) < 0)
                pf_check(-1);
            std::string text;
            text.reserve((size_t)nev * 48);
            char buf[96];
            for (int64_t i = 0; i < nev; ++i) {
                // out << eventcode << x + start_position - 1 << t << from_pop << to_pop, fixed, one decimal
                int len = snprintf(buf, sizeof buf, "%c\t%.1f\t%.1f\t%d\t%d\t", kind[i] == 0 ? 'R' : (kind[i] == 1 ? 'C' : 'M'),
                                   xs[i] + P.start_position - 1, ts[i], from[i], to[i]);
                text.append(buf, (size_t)len);
                // print_descendants (descendants.hpp:51-65)
                const uint32_t m = ds[i];
                if (m == 0) text += '0';
                else
                    for (int b = 0; (m >> b) != 0; ++b) text += ((m >> b) & 1u) ? '1' : '0';
                text += '\n';
            }
            gzFile gz = gzopen(P.trees_path.c_str(), "wb");
            if (gz) { gzwrite(gz, text.data(), (unsigned)text.size()); gzclose(gz); }
        }
        report_counts(P, M0, packed, 1.0);
    }
}

// pfARG_core (smcsmc.cpp:278-401): one E-step over one chunk
static void pfARG_core(PfParam& P, const HostModel& M0, int device, const ChunkJob* job = nullptr) {
    ChunkFilter F;
    open_filter(F, P, M0, device, job);
    std::vector<ChunkFilter*> one{&F};
    run_filters(one, true);
    close_filter(F, M0, job);
}

// log_counts + reset_model_parameters (count.cpp:44-158, 267-352) on the statistics of one E-step summed over `chunks`
// chunks: the rows of <prefix>.out and the in-binary M-step.  Every chunk brings the prior pseudo-counts of
// CountModel::init (count.cpp:161-227), as each chunk's own .out does when the front-end adds them up (model.py:1176-1184).
static void report_counts(PfParam& P, const HostModel& M0, const std::vector<double>& packed, double chunks) {
    const double K = chunks;
    // log_counts (count.cpp:66-158) with the prior pseudo-counts of init_coal_and_recomb (count.cpp:161-193)
    HostModel& M = P.model;
    const int E = (int)M.change_times.size();
    const int NP = M.npop;
    const double* tail = &packed[packed.size() - 4];
    const size_t EP = (size_t)E * NP;
    const double* cc = &packed[0]; const double* co = &packed[EP]; const double* cw = &packed[2 * EP];
    const double* rc = &packed[3 * EP]; const double* ro = &packed[3 * EP + E]; const double* rw = &packed[3 * EP + 2 * E];
    auto epoch_end = [&](int e) { return e == E - 1 ? 1e+99 : M.change_times[e + 1]; };
    // reset_model_parameters (count.cpp:44-63, forced at the end of every E-step, smcsmc.cpp:385-386): the M-step
    // of the in-binary EM.  Totals include the prior pseudo-counts of CountModel::init, which come from the
    // model the CountModel was constructed with (the initial one, smcsmc.cpp:77).
    {
        clog << " MODEL IS RESET at base " << M.loci_length << endl;
        double ro_ = 0, rc_ = 0, rw_ = 0;
        for (int e = 0; e < E; ++e) { ro_ += ro[e] + K; rc_ += rc[e] + K * M0.recombination_rate; rw_ += rw[e] + K; }
        std::vector<std::vector<double>> new_sizes = M.pop_sizes, new_mig = M.mig_rates;
        const double new_rho = rc_ / ro_;                                     // reset_recomb_rate, count.cpp:296-313
        clog << " Setting recombination rate to " << new_rho << " ( " << rc_ << " / " << ro_ << "; post-lag ESS "
             << 1.0 / (rw_ / ro_) << " )" << endl;
        for (int e = 0; e < E; ++e)                                           // reset_Ne, count.cpp:267-293
            for (int a = 0; a < NP; ++a) {
                const size_t k = (size_t)e * NP + a;
                double opp = co[k] + K, count = cc[k] + K / (2.0 * M0.pop_sizes[e][a]), weight = cw[k] + K;
                double pop_size = 1.0 / (2.0 * (count / opp));
                if (P.cap_sizes && pop_size >= P.size_cap) pop_size = P.size_cap;
                new_sizes[e][a] = pop_size;
                clog << " Setting size of population " << a << " @ " << setw(8) << M.change_times[e] << " to " << setw(8)
                     << pop_size << " ( 0.5 * " << opp << " / " << count << "; post-lag ESS " << 1.0 / (weight / opp)
                     << " )" << endl;
            }
        if (NP > 1) {                                                         // reset_mig_rate, count.cpp:327-352
            const double* mc = &packed[3 * EP + 3 * E]; const double* mo = mc + EP * NP;
            for (int e = 0; e < E; ++e)
                for (int a = 0; a < NP; ++a)
                    for (int b = 0; b < NP; ++b)
                        if (a != b) {
                            const size_t k = (size_t)e * NP + a;
                            new_mig[e][(size_t)a * NP + b] = (mc[k * NP + b] + K * M0.mig_rates[e][(size_t)a * NP + b]) / (mo[k] + K);
                        }
        }
        P.next_sizes = new_sizes; P.next_mig = new_mig; P.next_rho = new_rho;
    }
    for (int e = 0; e < E; ++e)
        for (int a = 0; a < NP; ++a) {
            const size_t k = (size_t)e * NP + a;
            P.write_out_row(P.em_iteration, e, M.change_times[e], epoch_end(e), "Coal", a, -1, co[k] + K,
                              cc[k] + K / (2.0 * M0.pop_sizes[e][a]), cw[k] + K);
        }
    // recombination is booked on population 0 only (count.cpp:534-539); the report sums the epochs (84-113)
    double ropp = 0, rcount = 0, rweight = 0;
    for (int e = 0; e < E; ++e) {
        ropp += ro[e] + K;
        rcount += rc[e] + K * M0.recombination_rate;
        rweight += rw[e] + K;
    }
    P.write_out_row(P.em_iteration, -1, 0.0, 1e+99, "Recomb", -1, -1, ropp, rcount, rweight);
    if (NP > 1) {
        // migration rows with the pseudo-counts of init_migr (count.cpp:196-227)
        const double* mc = &packed[3 * EP + 3 * E]; const double* mo = mc + EP * NP; const double* mw = mo + EP;
        for (int e = 0; e < E; ++e)
            for (int a = 0; a < NP; ++a)
                for (int b = 0; b < NP; ++b)
                    if (a != b) {
                        const size_t k = (size_t)e * NP + a;
                        P.write_out_row(P.em_iteration, e, M.change_times[e], epoch_end(e), "Migr", a, b, mo[k] + K,
                                          mc[k * NP + b] + K * M0.mig_rates[e][(size_t)a * NP + b], mw[k] + K);
                    }
    }
    double dopp = tail[0], dcount = tail[1], nres = tail[2], logl = tail[3];
    P.write_out_row(P.em_iteration, -1, 0.0, 1e+99, "Delay", -1, -1, dopp, dcount / (double)P.particles, dopp);
    P.write_out_row(P.em_iteration, -1, 0.0, 1e+99, "Resamp", -1, -1, dopp, nres, dopp);
    P.write_out_row(P.em_iteration, -1, 0, 1e+99, "LogL", -1, -1, 1.0, logl, 1.0);   // smcsmc.cpp:391
    clog << " Estimated log likelihood: " << logl << endl;
}

// ---- several chunks in one process -------------------------------------------------------------------------------
// The data-parallel layer of the reference is one OS process per chromosome chunk and a sum of the chunks' .out files in
// Python (smcsmc/model.py:563-662, 1050-1100, 1176-1184).  Here: K chunks of the window are dealt to R host threads
// ("ranks", rank r takes chunks r, r + R, ...), rank r drives device r mod G.  A rank filters ITS chunks side by side:
// all of them from a fresh prior (chunks are independent, smcsmc.cpp:296) through one kernel launch per row whose grid
// covers them (pf_run_many) when the single-launch row pipeline applies, one after the other otherwise.  Then ONE
// all-gather of the packed statistics (mgpu.cpp: RCCL over xGMI when every rank has its own device) and a sum in chunk
// order on every rank -- bit-identical for any R and G.
struct ChunkPlan {                      // what stays the same over the EM iterations of a run
    int K = 0, R = 0, G = 0, slots = 0;
    size_t LEN = 0;
    std::vector<int> device_of_rank;
    std::vector<long long> first;                       // first[c] .. first[c + 1]: the bases of chunk c
    std::vector<std::unique_ptr<Segment>> tables;       // the chunks' row tables, read and cut once (row cap of the initial model, pfparam.cpp:364)
    std::unique_ptr<CountAllGather> exchange;
};

static void plan_chunks(ChunkPlan& C, PfParam& P) {
    if (P.record_trees) throw Unsupported("-arg together with -chunks / -ranks (the tree dump is that of one filter)");
    if (P.write_resample) throw Unsupported("-record_ess together with -chunks / -ranks (one .resample file per filter)");
    C.K = P.chunks;
    C.G = P.devices > 0 ? P.devices : visible_devices();
    if (C.G < 1) throw std::runtime_error("no HIP device available (there is no CPU fallback)");
    if (C.G > visible_devices()) throw std::runtime_error("-devices exceeds the devices visible to this process");
    C.R = P.ranks > 0 ? P.ranks : std::min(C.G, C.K);
    if (C.R > C.K) C.R = C.K;
    const std::string transport = P.reduce_transport.empty() ? (C.R <= C.G ? "rccl" : "host") : P.reduce_transport;
    C.device_of_rank.resize(C.R);
    for (int r = 0; r < C.R; ++r) C.device_of_rank[r] = r % C.G;
    const int E = (int)P.model.change_times.size();
    C.LEN = (size_t)PF_COUNTS_LEN2(E, P.model.npop);
    C.slots = (C.K + C.R - 1) / C.R;                    // chunks per rank (the last ranks may hold one fewer)
    C.exchange.reset(new CountAllGather(C.R, C.device_of_rank, (size_t)C.slots * C.LEN, transport));
    const double L = P.model.loci_length;
    C.first.resize(C.K + 1);
    for (int c = 0; c <= C.K; ++c) C.first[c] = (long long)P.start_position + (long long)std::floor(L * c / C.K);
    const double cap = P.segment_cap();                 // from the initial recombination rate, once
    for (int c = 0; c < C.K; ++c)
        C.tables.emplace_back(new Segment(P.seg_path, P.nsam, (double)(C.first[c + 1] - C.first[c]), P.nodata_theta, C.first[c], cap));
}

static void run_chunks(ChunkPlan& C, PfParam& P, const HostModel& M0) {
    const int K = C.K, R = C.R, slots = C.slots;
    const size_t LEN = C.LEN;
    clog << " " << K << " chunks on " << R << " rank(s), " << C.G << " device(s); statistics exchanged by " << C.exchange->transport() << endl;
    std::vector<std::vector<double>> mine(R, std::vector<double>((size_t)slots * LEN, 0.0));
    std::vector<std::vector<double>> all(R, std::vector<double>((size_t)R * slots * LEN, 0.0));
    std::vector<std::string> failure(R);
    // the chunks of a rank go through one launch per row when the row pipeline applies to them (include/smcsmc_pf.h, pf_run_many)
    const bool lockstep = P.model.npop == 1 && P.model.nsam <= 8 && P.apf_level == 0;
    auto rank_main = [&](int r) {
        bool entered = false;
        try {
            std::vector<int> my_chunks;
            for (int c = r; c < K; c += R) my_chunks.push_back(c);
            std::vector<PfParam> params(my_chunks.size(), P);          // same flags and model, each its own piece of the data
            std::vector<ChunkFilter> filters(my_chunks.size());
            std::vector<ChunkJob> jobs(my_chunks.size());
            std::vector<std::vector<double>> packed(my_chunks.size());
            std::vector<double> survival;
            for (size_t k = 0; k < my_chunks.size(); ++k) {
                const int c = my_chunks[k];
                params[k].model.loci_length = (double)(C.first[c + 1] - C.first[c]);
                params[k].start_position = (double)C.first[c];
                params[k].segments = C.tables[c].get();
                // the chunk's own local recombination map, as a chunk process of the front-end would leave it
                // (<out>/emiterI/chunkC.recomb.gz there; model.py:1057-1092)
                params[k].recomb_map_path = P.recomb_map_path.substr(0, P.recomb_map_path.size() - std::string(".recomb.gz").size()) +
                                            ".chunk" + std::to_string(c) + ".recomb.gz";
                if (P.em_iteration == 0) remove(params[k].recomb_map_path.c_str());
                jobs[k].survival = &survival; jobs[k].survival_out = &survival; jobs[k].packed_out = &packed[k]; jobs[k].seed_offset = (uint64_t)c;
                // Six or more chunks side by side on a device: count_wgs set (to the most a column can have, one per 256 particles) makes
                // the library taper the columns of the young epochs and trim its ledger workgroups (8 chunks of the C3 shape: 1.17e5
                // segments/s against 1.09e5, 12 chunks 1.29e5 against 1.11e5; profiles/round4/wg_trace.md).  The sums of a chunk are grouped by workgroup: a run that must give the same bits
                // whatever the number of ranks pins the width with -count_wgs.
                jobs[k].count_wgs = P.count_wgs > 0 ? P.count_wgs : ((lockstep && my_chunks.size() >= 6) ? (int)((P.particles + 255) / 256) : 0);
            }
            // Side by side in groups: as many of the rank's chunks as the device has memory for are opened together (every
            // filter holds its own rings: the event log alone is Np x 16 384 records by default) and go through one launch per
            // row when the library says the group can (pf_can_run_many: the row pipeline applies to all of them -- one
            // population, at most 8 haplotypes, no look-ahead, Np <= 131 072 -- and they share their shape); otherwise, and
            // whenever a group comes down to one chunk, one after the other.
            size_t k0 = 0;
            while (k0 < my_chunks.size()) {
                std::vector<ChunkFilter*> group;
                size_t k1 = k0;
                while (k1 < my_chunks.size() && (k1 == k0 || lockstep)) {
                    try {
                        open_filter(filters[k1], params[k1], M0, C.device_of_rank[r], &jobs[k1]);
                    } catch (const std::exception& e) {
                        if (k1 == k0 || std::string(e.what()).find("memory") == std::string::npos) throw;
                        clog << " rank " << r << ": " << (k1 - k0) << " chunks fill the device; the rest follow" << endl;
                        break;                               // no room for another filter beside the open ones: run those first
                    }
                    group.push_back(&filters[k1]);
                    ++k1;
                }
                std::vector<pf_handle*> hs;
                for (ChunkFilter* F : group) hs.push_back(F->h);
                if (group.size() > 1 && pf_can_run_many(hs.data(), (int32_t)hs.size())) {
                    run_filters(group, r == 0);
                } else {
                    for (ChunkFilter* F : group) { std::vector<ChunkFilter*> one{F}; run_filters(one, r == 0); }
                }
                for (size_t k = k0; k < k1; ++k) { close_filter(filters[k], M0, &jobs[k]); release_filter(filters[k]); }
                k0 = k1;
            }
            for (size_t k = 0; k < my_chunks.size(); ++k) std::copy(packed[k].begin(), packed[k].end(), mine[r].begin() + k * LEN);
            entered = true;
            C.exchange->all_gather(r, mine[r].data(), all[r].data());
        } catch (const std::exception& e) {
            failure[r] = e.what();
        } catch (...) {
            failure[r] = "unknown failure";
        }
        if (!failure[r].empty() && !entered) {
            // the other ranks wait at the exchange: contribute zeros so that they can leave and report.  (A rank that
            // failed inside the exchange has already taken part in it: calling it again would leave an unmatched collective.)
            try { C.exchange->all_gather(r, mine[r].data(), all[r].data()); } catch (...) {}
        }
    };
    std::vector<std::thread> pool;
    for (int r = 1; r < R; ++r) pool.emplace_back(rank_main, r);
    rank_main(0);
    for (auto& t : pool) t.join();
    for (int r = 0; r < R; ++r) if (!failure[r].empty()) throw std::runtime_error("chunk rank " + std::to_string(r) + ": " + failure[r]);
    // every rank now holds every chunk's statistics: chunk c sits in rank c mod R, slot c / R.  Sum in chunk order.
    std::vector<double> total(LEN, 0.0);
    for (int c = 0; c < K; ++c) {
        const double* q = &all[0][((size_t)(c % R) * slots + (size_t)(c / R)) * LEN];
        for (size_t k = 0; k < LEN; ++k) total[k] += q[k];
    }
    for (int r = 1; r < R; ++r)                                   // all ranks received the same bytes
        if (memcmp(all[r].data(), all[0].data(), all[0].size() * sizeof(double)) != 0)
            throw std::runtime_error("the ranks disagree on the gathered statistics");
    report_counts(P, M0, total, (double)K);
}

int main(int argc, char* argv[]) {
    try {
        PfParam P;
        P.parse(argc, argv);
        if (P.version()) { P.print_version(&std::cout); return EXIT_SUCCESS; }
        if (P.help()) { P.print_help(); return EXIT_SUCCESS; }
        if (P.dump_model) { dump_model_json(P); return EXIT_SUCCESS; }
        if (P.dump_lookahead) { dump_lookahead_json(P); return EXIT_SUCCESS; }
        if (P.dump_segments) { dump_segments_json(P); return EXIT_SUCCESS; }
        P.write_out_header();
        const HostModel initial_model = P.model;        // what CountModel is constructed from (smcsmc.cpp:77)
        ChunkPlan chunk_plan;
        if (P.chunks > 1 || P.ranks > 1) plan_chunks(chunk_plan, P);
        for (int i = 0; i <= P.em_iterations; i++) {
            clog << "EM step " << i << endl;
            int device = 0;
            if (const char* d = getenv("SMCSMC_DEVICE")) device = atoi(d);      // which device a single-chunk run uses
            if (P.chunks > 1 || P.ranks > 1) run_chunks(chunk_plan, P, initial_model);
            else pfARG_core(P, initial_model, device);
            // the model the next E-step runs under (Model::addPopulationSize / addMigrationRate / setRecombinationRate)
            P.model.pop_sizes = P.next_sizes; P.model.mig_rates = P.next_mig; P.model.recombination_rate = P.next_rho;
            P.em_iteration++;
            clog << "End of EM step " << i << endl;
        }
        return P.log();
    } catch (const exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;   // smcsmc.cpp:99-102
        return EXIT_FAILURE;
    }
}
