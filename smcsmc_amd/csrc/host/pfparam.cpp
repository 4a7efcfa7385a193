// smcsmc_amd/csrc/host/pfparam.cpp -- flag system, model tables, .out/.log writers of the drop-in binary.
// Behaviour follows /root/reference/src/pfparam.cpp (line references inline); the scrm-style part of the
// command line is parsed by HostModel (the scrm fork itself is not available): only the
// flags the Python front-end emits (smcsmc/populationmodels.py:300-437) are understood.
#include "smcsmc_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>

using namespace std;

#ifndef SMCSMC_VERSION
#define SMCSMC_VERSION "smcsmc_amd-0.1.0"
#endif

// ------------------------------------------------------------------ helpers
static bool is_flag(const string& s) {
    // a token starting with '-' followed by a letter is an option (negative numbers are operands)
    return s.size() >= 2 && s[0] == '-' && (isalpha((unsigned char)s[1]) || s[1] == '@');
}

template <class T>
static T convert(const string& flag, const string& arg) {
    T value;
    std::stringstream ss(arg);
    ss >> value;
    if (ss.fail() || !ss.eof()) throw WrongType(arg);
    (void)flag;
    return value;
}

// ------------------------------------------------------------------ HostModel
namespace {
struct Change {
    std::vector<double> size;     // per pop, NaN = inherit
    std::vector<double> mig;      // P*P per generation, NaN = inherit
    std::vector<double> single;   // P*P
    std::vector<double> ccount;   // per pop: variational-Bayes coalescence event count (-vb), NaN = inherit
    std::vector<double> mcount;   // P*P: migration event counts, NaN = inherit
};
}  // namespace

void HostModel::parse(const std::vector<std::string>& tok) {
    // pass 1: population structure (-I) and -N0 must be known before rates are scaled
    size_t i = 0;
    double theta = -1, R = -1;
    bool have_t = false, have_r = false;
    for (size_t k = 0; k < tok.size(); ++k) {
        if (tok[k] == "-I") {
            if (k + 1 >= tok.size()) throw NotEnoughArg("-I");
            npop = convert<int>("-I", tok[k + 1]);
        }
    }
    const int P = npop;
    std::map<double, Change> changes;
    auto at = [&](double t_gen) -> Change& {
        Change& c = changes[t_gen];
        if (c.size.empty()) {
            c.size.assign(P, NAN);
            c.mig.assign((size_t)P * P, NAN);
            c.single.assign((size_t)P * P, 0.0);
            c.ccount.assign(P, NAN);
            c.mcount.assign((size_t)P * P, NAN);
        }
        return c;
    };
    at(0.0);
    auto need = [&](const string& flag, size_t k) {
        if (i + k >= tok.size()) throw NotEnoughArg(flag);
    };
    auto operand_follows = [&]() { return i + 1 < tok.size() && !is_flag(tok[i + 1]); };
    sample_pops.assign(nsam, 0);
    while (i < tok.size()) {
        const string f = tok[i];
        if (f == "-N0") {
            need(f, 1); N0 = convert<double>(f, tok[++i]);
        } else if (f == "-t") {
            need(f, 1); theta = convert<double>(f, tok[++i]); have_t = true;
        } else if (f == "-r") {
            need(f, 2); R = convert<double>(f, tok[++i]); loci_length = convert<double>(f, tok[++i]); have_r = true;
        } else if (f == "-I") {
            need(f, 1 + (size_t)P); ++i;
            int idx = 0, total = 0;
            for (int p = 0; p < P; ++p) {
                int c = convert<int>(f, tok[++i]);
                total += c;
                for (int k = 0; k < c && idx < nsam; ++k) sample_pops[idx++] = p;
            }
            if (total != nsam) throw InvalidInput("Sample sizes in -I do not add up to -nsam");
            if (operand_follows()) {   // optional symmetric migration rate
                double M = convert<double>(f, tok[++i]);
                Change& c = at(0.0);
                for (int a = 0; a < P; ++a)
                    for (int b = 0; b < P; ++b)
                        if (a != b) c.mig[(size_t)a * P + b] = M / (P - 1) / (4 * N0);
            }
        } else if (f == "-eN") {
            need(f, 2); double t = convert<double>(f, tok[++i]) * 4 * N0; double x = convert<double>(f, tok[++i]);
            double cnt = NAN;
            if (vb) { need(f, 1); cnt = convert<double>(f, tok[++i]); }
            Change& c = at(t);
            for (int p = 0; p < P; ++p) { c.size[p] = x * N0; c.ccount[p] = cnt; }
        } else if (f == "-en") {
            need(f, 3); double t = convert<double>(f, tok[++i]) * 4 * N0; int p = convert<int>(f, tok[++i]);
            double x = convert<double>(f, tok[++i]);
            double cnt = NAN;
            if (vb) { need(f, 1); cnt = convert<double>(f, tok[++i]); }
            if (p < 1 || p > P) throw InvalidInput("Population index out of range in -en");
            at(t).size[p - 1] = x * N0;
            at(t).ccount[p - 1] = cnt;
        } else if (f == "-eM") {
            need(f, 2); double t = convert<double>(f, tok[++i]) * 4 * N0; double M = convert<double>(f, tok[++i]);
            double cnt = NAN;
            if (vb) { need(f, 1); cnt = convert<double>(f, tok[++i]); }
            Change& c = at(t);
            for (int a = 0; a < P; ++a)
                for (int b = 0; b < P; ++b)
                    if (a != b) {
                        c.mig[(size_t)a * P + b] = (P > 1 ? M / (P - 1) : 0.0) / (4 * N0);
                        // [INFERRED] the event count of the symmetric form is the total over the matrix, as the
                        // front-end writes it (populationmodels.py:362-376): spread like the rate
                        c.mcount[(size_t)a * P + b] = cnt / ((double)P * P);
                    }
        } else if (f == "-em") {
            need(f, 4); double t = convert<double>(f, tok[++i]) * 4 * N0; int a = convert<int>(f, tok[++i]);
            int b = convert<int>(f, tok[++i]); double M = convert<double>(f, tok[++i]);
            double cnt = NAN;
            if (vb) { need(f, 1); cnt = convert<double>(f, tok[++i]); }
            if (a < 1 || a > P || b < 1 || b > P) throw InvalidInput("Population index out of range in -em");
            at(t).mig[(size_t)(a - 1) * P + (b - 1)] = M / (4 * N0);
            at(t).mcount[(size_t)(a - 1) * P + (b - 1)] = cnt;
        } else if (f == "-ema") {
            need(f, 1 + (size_t)P * P * (vb ? 2 : 1)); double t = convert<double>(f, tok[++i]) * 4 * N0;
            Change& c = at(t);
            for (int a = 0; a < P; ++a)
                for (int b = 0; b < P; ++b) {
                    const string& v = tok[++i];
                    double cnt = NAN;
                    if (vb) cnt = convert<double>(f, tok[++i]);
                    if (a != b) {
                        c.mig[(size_t)a * P + b] = (v == "x" ? 0.0 : convert<double>(f, v)) / (4 * N0);
                        c.mcount[(size_t)a * P + b] = cnt;
                    }
                }
        } else if (f == "-ej") {
            need(f, 3); double t = convert<double>(f, tok[++i]) * 4 * N0; int a = convert<int>(f, tok[++i]);
            int b = convert<int>(f, tok[++i]);
            if (a < 1 || a > P || b < 1 || b > P) throw InvalidInput("Population index out of range in -ej");
            Change& c = at(t);
            c.single[(size_t)(a - 1) * P + (b - 1)] = 1.0;
            // scrm -ej: every line of population a moves to b, and no line may enter a from then on
            for (int k = 0; k < P; ++k)
                if (k != a - 1) { c.mig[(size_t)k * P + (a - 1)] = 0.0; c.mig[(size_t)(a - 1) * P + k] = 0.0; }
        } else if (f == "-seed") {
            need(f, 1);
            uint64_t sd = 0; int k = 0;
            while (k < 3 && operand_follows()) { sd = sd * 1000003ULL + (uint64_t)convert<long long>(f, tok[++i]); ++k; }
            if (k == 0) throw NotEnoughArg(f);
            seed = sd; seed_set = true;
        } else if (f == "-l") {
            need(f, 1); string v = tok[++i];
            if (!v.empty() && v.back() == 'r') throw Unsupported("-l <n>r (recombination-count window)");
            window_length_seq = convert<double>(f, v);
            if (window_length_seq != 0) throw Unsupported("-l " + v + " (only the SMC' window -l 0 is implemented)");
        } else if (f == "-vb") {
            vb = true;
        } else if (f == "-bias_heights") {
            while (operand_follows()) bias_heights.push_back(convert<double>(f, tok[++i]));
        } else if (f == "-bias_strengths") {
            while (operand_follows()) bias_strengths.push_back(convert<double>(f, tok[++i]));
        } else if (f == "-eI") {
            throw Unsupported("-eI (ancient samples)");
        } else {
            throw UnknowArg(f);
        }
        ++i;
    }
    if (!have_r) throw std::invalid_argument("The option -r is required");     // pfparam.cpp:266-270
    if (!have_t) throw std::invalid_argument("The option -t is required");     // pfparam.cpp:258-263
    // Model::setMutationRate / setRecombinationRate (scrm): scaled by 4*N0, per locus
    mutation_rate = theta / (4 * N0) / loci_length;
    if (loci_length <= 1) throw InvalidInput("Locus length must be larger than 1");
    recombination_rate = R / (4 * N0) / (loci_length - 1);
    // Model::finalize: fill unset values forward in time
    change_times.clear(); pop_sizes.clear(); mig_rates.clear(); single_mig.clear(); coal_counts.clear(); mig_counts.clear();
    std::vector<double> cur_size(P, N0), cur_mig((size_t)P * P, 0.0);
    std::vector<double> cur_cc(P, 1e10), cur_mc((size_t)P * P, 1e10);      // populationmodels.py:259-268 defaults
    for (auto& kv : changes) {
        Change& c = kv.second;
        for (int p = 0; p < P; ++p) if (!std::isnan(c.size[p])) cur_size[p] = c.size[p];
        for (size_t k = 0; k < c.mig.size(); ++k) if (!std::isnan(c.mig[k])) cur_mig[k] = c.mig[k];
        for (int p = 0; p < P; ++p) if (!std::isnan(c.ccount[p])) cur_cc[p] = c.ccount[p];
        for (size_t k = 0; k < c.mcount.size(); ++k) if (!std::isnan(c.mcount[k])) cur_mc[k] = c.mcount[k];
        coal_counts.push_back(cur_cc);
        mig_counts.push_back(cur_mc);
        change_times.push_back(kv.first);
        pop_sizes.push_back(cur_size);
        mig_rates.push_back(cur_mig);
        single_mig.push_back(c.single);
    }
}

void HostModel::finalize() {}

// ------------------------------------------------------------------ driver options
// The flags the binary itself consumes are one table: name, operand kind, help text, what to do with the operand.
// Everything that is not in the table belongs to the scrm-style model description and is handed to HostModel::parse
// (the reference's two-stage scheme, pfparam.cpp:63-169).  Names, defaults, ranges and user-visible strings are the
// contract; -help is printed from the same table.
namespace {

// "a" or "a-b" (closed, 0-based)
EpochRange epoch_range(const std::string& flag, const std::string& text) {
    const size_t dash = text.find('-', 1);
    EpochRange r;
    r.first = convert<int>(flag, text.substr(0, dash));
    r.last = dash == std::string::npos ? r.first : convert<int>(flag, text.substr(dash + 1));
    return r;
}

struct DriverOption {
    const char* name;
    const char* operand;     // "INT" / "FLT" / "STR", or "" for a switch
    const char* group;       // help section; nullptr: not listed
    const char* help;
    std::function<void(PfParam&, const std::string&)> apply;
};

const std::vector<DriverOption>& driver_options() {
    static const std::vector<DriverOption> table = {
        {"-Np", "INT", "Options", "Number of particles [ 100 ]",
         [](PfParam& p, const std::string& v) { p.particles = convert<size_t>("-Np", v); }},
        {"-nsam", "INT", "Options", "Number of haplotypes [ 2 ]",
         [](PfParam& p, const std::string& v) { p.nsam = convert<size_t>("-nsam", v); }},
        {"-seg", "STR", "Options", "Data file in seg format [ Chrom1.seg ]", [](PfParam& p, const std::string& v) { p.seg_path = v; }},
        {"-o", "STR", "Options", "Prefix for output files", [](PfParam& p, const std::string& v) { p.out_prefix = v; }},
        {"-EM", "INT", "Options", "EM iterations [ 0 ]",
         [](PfParam& p, const std::string& v) { p.em_iterations = convert<int>("-EM", v); }},
        {"-startpos", "INT", "Options", "First nucleotide position to analyze [ 1 ]",
         [](PfParam& p, const std::string& v) { p.start_position = convert<double>("-startpos", v); if (p.start_position < 1) throw OutOfRange("-startpos", v); }},
        {"-apf", "INT", "Options", "Use auxiliary particle filter [ 0 ]; 1 singletons, 2 + doubletons, 3 + splits, 4 n-choose-k split factor",
         [](PfParam& p, const std::string& v) { p.apf_level = convert<int>("-apf", v); if (p.apf_level < 0 || p.apf_level > 4) throw OutOfRange("-apf", v); }},
        {"-guide", "STR", "Options", "Recombination guide file (locus, size, rate, relative rate per sample)",
         [](PfParam& p, const std::string& v) { p.guide_path = v; }},
        {"-arg", "", "Options", "Sample a posterior ARG; write *.trees.gz", [](PfParam& p, const std::string&) { p.record_trees = true; }},
        {"-log", "", "Options", "Generate *.log file", [](PfParam& p, const std::string&) { p.write_log_file = true; }},
        {"-v", "", "Options", "Display timestamp and versions", [](PfParam& p, const std::string&) { p.want_version = true; }},
        {"-version", "", nullptr, "", [](PfParam& p, const std::string&) { p.want_version = true; }},
        {"-h", "", nullptr, "", [](PfParam& p, const std::string&) { p.want_help = true; }},
        {"-help", "", nullptr, "", [](PfParam& p, const std::string&) { p.want_help = true; }},
        {"-dephase", "", "Inference tuning", "Dephase heterozygous sites [ false ]", [](PfParam& p, const std::string&) { p.dephase = true; }},
        {"-calibrate_lag", "FLT", "Inference tuning", "Lag before extracting events (multiple of survival time) [ 2 ]",
         [](PfParam& p, const std::string& v) { p.calibrate_lag = true; p.lag_fraction = convert<double>("-calibrate_lag", v); if (p.lag_fraction < 0.0) throw OutOfRange("-calibrate_lag", v); }},
        {"-lag", "FLT", "Inference tuning", "Constant lag (bp); disables calibration",
         [](PfParam& p, const std::string& v) { p.lag = convert<double>("-lag", v); p.calibrate_lag = false; }},
        {"-delay", "FLT", "Inference tuning", "Delay of importance weights (multiple of survival time) [ 0.5 ]",
         [](PfParam& p, const std::string& v) { p.delay = convert<double>("-delay", v); if (p.delay < 0.0) throw OutOfRange("-delay", v); }},
        {"-delay_coal", "", "Inference tuning", "Delay by the coalescence height instead of the recombination height",
         [](PfParam& p, const std::string&) { p.delay_type = 1; }},
        {"-delay_migr", "", "Inference tuning", "Delay by the first coalescence or migration height",
         [](PfParam& p, const std::string&) { p.delay_type = 2; }},
        // not a reference flag: the block that applies the importance factor of an event above the focused band at once
        // (particle.cpp:878-885) is left out, as on the branch the reference's two-population bands were calibrated on
        // ("2b3a_wo_apply_immediately_hack", test_two_pops.py:50)
        {"-delay_all", "", "Inference tuning", "Delay every importance factor of focused sampling, also above the focused band",
         [](PfParam& p, const std::string&) { p.delay_all = true; }},
        {"-tmax", "FLT", "Inference tuning", "Maximum tree height, in unit of 4N0 [ 2 ]",
         [](PfParam& p, const std::string& v) { p.tmax = convert<double>("-tmax", v); }},
        {"-p", "STR", "Inference tuning", "Pattern of time segments, e.g. 1*3+15*4+1",
         [](PfParam& p, const std::string& v) { p.pattern = v; }},
        {"-ESS", "FLT", "Inference tuning", "Fractional ESS threshold for resampling [ 0.5 ]",
         [](PfParam& p, const std::string& v) { p.ess_fraction = convert<double>("-ESS", v); p.ess_is_default = false; if (p.ess_fraction > 1.0 || p.ess_fraction < 0.0) throw OutOfRange("-ESS", v); }},
        {"-xr", "INT", "Inference tuning", "Epoch or epoch range to exclude from recombination EM (0-based, closed)",
         [](PfParam& p, const std::string& v) { p.exclude_recomb.push_back(epoch_range("-xr", v)); }},
        {"-xc", "INT", "Inference tuning", "Epoch or epoch range (e.g. 0-10) to exclude from coalescent/migration EM",
         [](PfParam& p, const std::string& v) { p.exclude_coalmigr.push_back(epoch_range("-xc", v)); }},
        {"-cap", "FLT", "Inference tuning", "Upper bound on effective population sizes in the in-binary M-step",
         [](PfParam& p, const std::string& v) { p.size_cap = convert<double>("-cap", v); p.cap_sizes = true; }},
        {"-ancestral_aware", "", "Inference tuning", "Ancestral allele is 0", [](PfParam& p, const std::string&) { p.ancestral_aware = true; }},
        {"-record_ess", "", "Inference tuning", "Generate *.resample file", [](PfParam& p, const std::string&) { p.write_resample = true; }},
        // not reference flags: several chunks of the window in one process, on one or several devices (main.cpp: run_chunks)
        {"-chunks", "INT", "Several chunks", "Cut the window into this many chunks, filtered independently; statistics summed in chunk order [ 1 ]",
         [](PfParam& p, const std::string& v) { p.chunks = convert<int>("-chunks", v); if (p.chunks < 1) throw OutOfRange("-chunks", v); }},
        {"-ranks", "INT", "Several chunks", "Host threads filtering chunks concurrently (rank r takes chunks r, r + ranks, ...) [ number of devices ]",
         [](PfParam& p, const std::string& v) { p.ranks = convert<int>("-ranks", v); if (p.ranks < 1) throw OutOfRange("-ranks", v); }},
        {"-devices", "INT", "Several chunks", "Devices to use (rank r drives device r mod devices) [ all visible ]",
         [](PfParam& p, const std::string& v) { p.devices = convert<int>("-devices", v); if (p.devices < 1) throw OutOfRange("-devices", v); }},
        {"-reduce", "STR", "Several chunks", "Exchange of the statistics between ranks: rccl (one device per rank) or host [ rccl when possible ]",
         [](PfParam& p, const std::string& v) { p.reduce_transport = v; }},
        // not a reference flag: room for migration events on one local tree (the reference's node list is unbounded)
        {"-migcap", "INT", "Several populations", "Migration events one local tree may hold; about 230 fit the LDS with 32 epochs [ 96 ]",
         [](PfParam& p, const std::string& v) { p.mig_cap = convert<int>("-migcap", v); if (p.mig_cap < 1) throw OutOfRange("-migcap", v); }},
        // not reference flags: room for pending delayed importance factors per particle (the reference's heap is unbounded,
        // particle.hpp:248), and what happens when it runs out
        {"-log_cap", "INT", "Inference tuning", "Event-log records kept per particle (a ring; one too few for the lags in force stops the run with a message) [ 16384 ]",
         [](PfParam& p, const std::string& v) { p.log_cap = convert<long long>("-log_cap", v); if (p.log_cap < 4) throw OutOfRange("-log_cap", v); }},
        {"-count_wgs", "INT", "Inference tuning", "Workgroups per epoch that share the lagged counting of a row (part of what makes two runs bit-identical) [ one per 256 particles; with six or more chunks per device fewer for the young epochs ]",
         [](PfParam& p, const std::string& v) { p.count_wgs = convert<int>("-count_wgs", v); if (p.count_wgs < 1) throw OutOfRange("-count_wgs", v); }},
        {"-delaycap", "INT", "Inference tuning", "Delayed importance factors a particle may have pending; one too many stops the run [ 128 ]",
         [](PfParam& p, const std::string& v) { p.delay_cap = convert<int>("-delaycap", v); if (p.delay_cap < 1) throw OutOfRange("-delaycap", v); }},
        {"-delay_evict", "", "Inference tuning", "A full store of delayed factors applies its earliest factor early instead of stopping (counted in the log)",
         [](PfParam& p, const std::string&) { p.delay_evict = true; }},
        // not a reference flag: keep recording events however far the next informative site is.  The surveyed reference stops
        // recording epoch e beyond half a lag from data (max_epoch_to_update, smcsmc.cpp:266-275), which on an all-missing
        // file leaves almost nothing to count; its no-data regression bands predate that rule
        {"-record_all", "", "Inference tuning", "Record events in every epoch at every row (no limit far from data)",
         [](PfParam& p, const std::string&) { p.record_all = true; }},
        // not reference flags: print what the host side made of the input, as JSON, and exit (used by the tests)
        {"-dumpmodel", "", nullptr, "", [](PfParam& p, const std::string&) { p.dump_model = true; }},
        {"-dumplookahead", "", nullptr, "", [](PfParam& p, const std::string&) { p.dump_lookahead = true; }},
        {"-dumpsegments", "", nullptr, "", [](PfParam& p, const std::string&) { p.dump_segments = true; }},
    };
    return table;
}

const DriverOption* find_option(const std::string& name) {
    static const std::map<std::string, const DriverOption*> index = [] {
        std::map<std::string, const DriverOption*> m;
        for (const DriverOption& o : driver_options()) m[o.name] = &o;
        return m;
    }();
    auto it = index.find(name);
    return it == index.end() ? nullptr : it->second;
}

}  // namespace

void PfParam::parse(int argc, char* argv[]) {
    if (argc <= 1) { want_help = true; return; }
    cmdline = argv[0];
    for (int k = 1; k < argc; ++k) cmdline += std::string(" ") + argv[k];
    nodata_theta = 1e-8 * 40000 * 2e7;          // no -seg: expected mutations of the reference's default locus (pfparam.cpp:195-198)
    for (int k = 1; k < argc; ++k) {
        const std::string token = argv[k];
        const DriverOption* opt = find_option(token);
        if (!opt) { model_tokens.push_back(token); continue; }      // part of the model description
        if (opt->operand[0] == '\0') { opt->apply(*this, ""); continue; }
        if (k + 1 >= argc) throw NotEnoughArg(token);
        opt->apply(*this, argv[++k]);
    }
    if (want_help || want_version) return;
    finalize();
    clog << "Command line:-" << endl << cmdline << endl;
}

// RecombinationBias::read_guide_file and operator>>(RecombBiasSegment) (pfparam.hpp:124-198): tab-separated
// `locus size recomb_rate 1 .. n`, 0-based, no gaps, plain or gzipped; the records that start inside the locus are
// the ones the model takes (set_model_rates, pfparam.hpp:202-212)
void PfParam::read_guide_file(const std::string& filename) {
    gzFile in = gzopen(filename.c_str(), "rb");          // reads plain text as well
    if (!in) {
        cout << "Problem opening file " << filename << endl;
        throw InvalidInput("Recombination guide file could not be opened.");
    }
    std::string text;
    char buf[1 << 16];
    int got;
    while ((got = gzread(in, buf, sizeof buf)) > 0) text.append(buf, (size_t)got);
    gzclose(in);
    std::istringstream in_file(text);
    std::string header, line, elt;
    std::getline(in_file, header);
    if (header.substr(0, 5) != "locus")
        throw InvalidInput("Expected header line (with columns 'locus', 'size', 'recomb_rate', '1', ...) in recombination guide file");
    long end = 0;
    while (std::getline(in_file, line)) {
        if (std::count(line.begin(), line.end(), ' ') > 0) {
            cerr << "Found spaces in recombination record; columns must be tab-separated" << endl;
            cerr << "Record: '" << line << "'" << endl;
            throw InvalidInput("Found spaces in recombination record");
        }
        long locus = 0, size = 0;
        double rate = 0;
        std::vector<double> leaf;
        std::stringstream iss(line);
        if (!std::getline(iss, elt, '\t')) throw InvalidInput("Parse error on 1st element");
        try {
            locus = std::stoi(elt);
            if (!std::getline(iss, elt, '\t')) throw InvalidInput("Parse error on 2nd element");
            size = std::stoi(elt);
            if (!std::getline(iss, elt, '\t')) throw InvalidInput("Parse error on 3rd element");
            rate = std::stod(elt);
            while (std::getline(iss, elt, '\t')) leaf.push_back(std::stod(elt));
        } catch (const InvalidInput&) {
            throw;
        } catch (...) {
            throw InvalidInput("Problem reading or parsing recombination guide file");
        }
        if (leaf.size() != nsam) {
            cerr << "Problem on record at position " << locus << " with " << leaf.size() << " leaf columns; expected "
                 << nsam << endl;
            throw InvalidInput("Did not find expected number of leaf columns");
        }
        if (locus != end) {
            cerr << "Problem on record at position " << locus << endl;
            throw InvalidInput("Did not get expected locus position (records should start at 0, and leave no gaps)");
        }
        end = locus + size;
        if ((double)locus < model.loci_length) {
            guide_positions.push_back((double)locus);
            guide_rates.push_back(rate);
            guide_leaf_rates.insert(guide_leaf_rates.end(), leaf.begin(), leaf.end());
        }
    }
    if (guide_positions.empty()) throw InvalidInput("Recombination guide file holds no records");
}

// The -p pattern of time segments (Pattern, pattern.cpp:33-163), PSMC style: "a*b" = a groups of b atomic intervals,
// a bare number = one group of that many.  The n atomic boundaries are t_i = 0.1 exp(i/(n-1) log(1 + 10 tmax)) - 0.1
// (i = 0..n-1, so t_0 = 0 and t_{n-1} = tmax); every group starts an epoch, written as "-eN <t/2> 1" with the six
// decimals of std::to_string.  Fewer than two atomic intervals: no epochs.
std::vector<std::string> expand_pattern(const std::string& pattern, double tmax) {
    const char* expr = pattern.c_str();
    auto number = [&]() -> size_t {
        if (!isdigit((unsigned char)*expr)) throw PatternDigitsExpected(std::string(expr));
        char* end_ptr;
        size_t res = (size_t)strtol(expr, &end_ptr, 10);
        expr = end_ptr;
        return res;
    };
    std::vector<size_t> groups, sizes;
    auto factor = [&]() -> size_t {
        size_t a = number();
        char op = *expr;
        if (op == '+' || op == '\0') { groups.push_back(1); sizes.push_back(a); return a; }
        if (op != '*') throw PatternTimesExpected(std::string(expr));
        ++expr;
        size_t b = number();
        groups.push_back(a); sizes.push_back(b);
        op = *expr;
        if (op != '+' && op != '\0') throw PatternAddsExpected(std::string(expr));
        return a * b;
    };
    size_t num_seg = factor();
    while (*expr) { ++expr; num_seg += factor(); }
    std::vector<std::string> out;
    if (num_seg < 2) return out;
    std::vector<double> t_i(num_seg);
    for (size_t i = 0; i < num_seg; ++i)
        t_i[i] = 0.1 * exp((double)i / (double)(num_seg - 1) * log(1 + 10 * tmax)) - 0.1;
    size_t index = 0;
    for (size_t g = 0; g < groups.size(); ++g)
        for (size_t i = 0; i < groups[g]; ++i) {
            out.push_back("-eN");
            out.push_back(std::to_string(t_i[index] / 2));
            out.push_back("1");
            index += sizes[g];
        }
    return out;
}

// pfparam.cpp:321-380
void PfParam::finalize() {
    out_path = out_prefix + ".out";
    log_path = out_prefix + ".log";
    recomb_map_path = out_prefix + ".recomb.gz";
    trees_path = out_prefix + ".trees.gz";
    resample_path = out_prefix + ".resample";
    if (!guide_path.empty() && apf_level > 0)
        throw std::invalid_argument("Recombination guiding and auxiliary particle filters cannot currently be used together");
    if (!dump_model && !dump_lookahead && !dump_segments) {
        remove(out_path.c_str());
        remove(log_path.c_str());
        remove(recomb_map_path.c_str());
        if (write_resample) remove(resample_path.c_str());
    }
    if (record_trees && !dump_model && !dump_lookahead && !dump_segments) remove(trees_path.c_str());
    if (!pattern.empty()) {
        // pfparam.cpp:292-295: the epochs of the pattern are appended to the scrm arguments
        for (const std::string& tok : expand_pattern(pattern, tmax)) model_tokens.push_back(tok);
    }
    model.nsam = (int)nsam;
    model.parse(model_tokens);
    if (!guide_path.empty()) {
        read_guide_file(guide_path);
    }
    if (model.change_times.back() >= tmax * 40000)
        throw std::invalid_argument("Problem: -tmax must be larger than bottom of final epoch");
    if (!model.bias_heights.empty() || !model.bias_strengths.empty()) {
        if (model.bias_strengths.size() != model.bias_heights.size() + 1)
            throw std::invalid_argument("-bias_strengths should have one more value than -bias_heights");
        if (model.bias_heights.size() > 8) throw Unsupported("more than 8 bias heights");
    }
    // which events are recorded per epoch: everything, minus the -xr / -xc ranges
    const int E = (int)model.change_times.size();
    record_mask.assign((size_t)E, RECORD_COALMIGR | RECORD_RECOMB);
    auto exclude = [&](const std::vector<EpochRange>& ranges, int bit) {
        for (const EpochRange& r : ranges) {
            if (r.last >= E) throw OutOfEpochRange(to_string(r.last), to_string(E - 1));
            for (int e = std::max(0, r.first); e <= r.last; ++e) record_mask[(size_t)e] &= ~bit;
        }
    };
    exclude(exclude_recomb, RECORD_RECOMB);
    exclude(exclude_coalmigr, RECORD_COALMIGR);
    int max_seg_len = (int)segment_cap();
    if (!dump_model && !(chunks > 1 || ranks > 1))      // several chunks: main.cpp reads one table per chunk instead
        segments = new Segment(seg_path, nsam, model.loci_length, nodata_theta,
                              (long long)start_position, max_seg_len);
}

double PfParam::segment_cap() const { return (double)(int)(row_cap_factor / (model.recombination_rate * 4 * model.N0)); }

// ------------------------------------------------------------------ writers
std::string format_double(double d, double scientific_bound, int precision) {   // pfparam.cpp:482-497
    const int field_length = 14;
    const double maxdouble = exp((field_length - precision - 1) * log(10.0));
    std::ostringstream o;
    if (d < maxdouble && (d > scientific_bound || d == 0.0)) o << setw(field_length) << fixed << setprecision(precision) << d;
    else o << setw(field_length) << scientific << setprecision(field_length - 7) << d;
    return o.str();
}

void PfParam::write_out_header() {   // pfparam.cpp:459-479
    ofstream f(out_path.c_str(), ios::binary);
    const int f1 = 6, f2 = 14;
    f << setw(f1) << "Iter" << " " << setw(f1) << "Epoch" << " " << setw(f2) << "Start" << " " << setw(f2) << "End" << " "
      << setw(f1) << "Type" << " " << setw(f1) << "From" << " " << setw(f1) << "To" << " " << setw(f2) << "Opp" << " "
      << setw(f2) << "Count" << " " << setw(f2) << "Rate" << " " << setw(f2) << "Ne" << " " << setw(f2) << "ESS" << endl;
}

void PfParam::write_out_row(size_t EMstep, int epoch, double epochBegin, double epochEnd, string eventType, int from_pop,
                              int to_pop, double opportunity, double count, double weight) {   // pfparam.cpp:500-527
    ofstream f(out_path.c_str(), ios::out | ios::app | ios::binary);
    const int f1 = 6;
    f << setw(f1) << EMstep << " " << setw(f1) << epoch << " " << format_double(epochBegin) << " " << format_double(epochEnd) << " "
      << setw(f1) << eventType << " " << setw(f1) << from_pop << " " << setw(f1) << to_pop << " " << format_double(opportunity)
      << " " << format_double(count) << " " << format_double(count / (opportunity + 1e-10)) << " "
      << format_double((eventType == "Coal") ? (opportunity + 1e-10) / (2.0 * count) : 0.0) << " "
      << format_double(1.0 / (weight / opportunity + 1e-10), 1.0, 3) << endl;
}

void PfParam::write_resample_row(double position, double ESS) const {   // pfparam.cpp:530-538
    if (!write_resample) return;
    ofstream f(resample_path.c_str(), ios::out | ios::app | ios::binary);
    f << (int)position << "\t" << ESS << endl;
}

void PfParam::print_version(std::ostream* o) {   // pfparam.cpp:588-592
    (*o) << "Program was compiled on: " << __DATE__ << endl;
    (*o) << "smcsmc version: " << SMCSMC_VERSION << endl;
    (*o) << "scrm version:   " << "none(hip-native-smc-prime)" << endl;
}

void PfParam::print_help() {
    cout << "smcsmc (MI355X build) -- particle filter for demographic inference -- Version " << SMCSMC_VERSION << endl;
    const char* section = nullptr;
    for (const DriverOption& o : driver_options()) {
        if (!o.group) continue;
        if (!section || strcmp(section, o.group) != 0) {
            cout << (section ? "\n" : "") << o.group << ":" << endl;
            section = o.group;
        }
        cout << setw(15) << o.name << setw(8) << (o.operand[0] ? o.operand : " ") << "  --  " << o.help << endl;
    }
    cout << endl << "Model (scrm-style): -N0 -t -r -I -eN -en -eM -em -ema -ej -seed -l 0 -vb -bias_heights -bias_strengths" << endl;
}

void PfParam::write_log_text(ostream* w) {   // pfparam.cpp:403-456
    (*w) << "###########################\n";
    (*w) << "#        smcsmc log       #\n";
    (*w) << "###########################\n";
    print_version(w);
    (*w) << "smcsmc parameters: \n";
    (*w) << "Segment Data file: " << (seg_path.empty() ? "empty" : seg_path.c_str()) << "\n";
    (*w) << "Recombination bias file: " << (guide_path.empty() ? "None" : guide_path.c_str()) << "\n";
    (*w) << setw(15) << " EM steps =" << setw(10) << em_iterations << "\n";
    if (lag > 0) (*w) << setw(15) << "Constant lag =" << setw(10) << lag << "\n";
    (*w) << setw(15) << "N =" << setw(10) << particles << "\n";
    (*w) << setw(15) << "ESS =" << setw(10) << ess_fraction;
    if (ess_is_default) (*w) << " (by default)";
    (*w) << "\n";
    (*w) << "scrm model parameters: \n";
    (*w) << setw(17) << "Extract window =" << setw(10) << model.window_length_seq << "\n";
    (*w) << setw(17) << "Sample size =" << setw(10) << model.nsam << "\n";
    (*w) << setw(17) << "Seq length =" << setw(10) << model.loci_length << "\n";
    (*w) << setw(17) << "mutation rate =" << setw(10) << model.mutation_rate << "\n";
    (*w) << setw(17) << "recomb rate =" << setw(10) << model.recombination_rate << "\n";
    (*w) << setw(17) << "Pop size (at Generation):\n";
    for (size_t e = 0; e < model.change_times.size(); e++) {
        (*w) << setw(3) << "(" << setw(8) << model.change_times[e] << " )";
        for (int p = 0; p < model.npop; p++) (*w) << " | " << setw(10) << model.pop_sizes[e][p];
        (*w) << "\n";
    }
    (*w) << "Out file is saved in file: " << out_path << "\n";
}

int PfParam::log() {   // pfparam.cpp:392-400
    if (write_log_file) {
        ofstream f(log_path.c_str(), ios::out | ios::app | ios::binary);
        write_log_text(&f);
    }
    write_log_text(&std::cout);
    return 0;
}
