// smcsmc_amd/csrc/host/smcsmc_host.hpp -- host side of the drop-in `smcsmc` binary.
//
// Mirrors the reference's driver layer (paths relative to /root/reference/src):
//   PfParam        pfparam.hpp:225-446, pfparam.cpp   (flag system, .out/.log writers)
//   HostModel      the part of the scrm fork's Param/Model that the front-end exercises
//                  (flags emitted by smcsmc/populationmodels.py:300-437)
//   Segment        segdata.hpp:86-177, segdata.cpp    (.seg reader)
// The compute path is the C-ABI of include/smcsmc_pf.h (HIP, gfx950); nothing here computes
// particle-filter results on the CPU.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

// ---- exception taxonomy (exception.hpp:32-51, pfparam.hpp:42-93, segdata.hpp:41-81) ----
struct InvalidInput : public std::exception {
    std::string src, reason, throwMsg;
    InvalidInput() : InvalidInput("") {}
    explicit InvalidInput(std::string s) : src("\033[1;31m" + s + "\033[0m"), reason(""), throwMsg(s) {}
    ~InvalidInput() throw() override {}
    const char* what() const noexcept override { return throwMsg.c_str(); }
};
struct NotEnoughArg : InvalidInput {
    explicit NotEnoughArg(std::string s) : InvalidInput(s) { reason = "Not enough parameters when parsing option: "; throwMsg = reason + src; }
};
struct UnknowArg : InvalidInput {
    explicit UnknowArg(std::string s) : InvalidInput(s) { reason = "Unknow option: "; throwMsg = reason + src; }
};
struct OutOfEpochRange : InvalidInput {
    OutOfEpochRange(std::string a, std::string b) : InvalidInput(a) {
        reason = "Problem: epochs specified in -xr/-xc options out of range: ";
        src = "\033[1;31m" + a + std::string(" is greater than ") + b + "\033[0m";
        throwMsg = reason + src;
    }
};
struct OutOfRange : InvalidInput {
    OutOfRange(std::string a, std::string b) : InvalidInput(a) {
        reason = "Flag \"";
        throwMsg = reason + src + std::string(" ") + b + std::string("\" out of range [0, 1].");
    }
};
struct WrongType : InvalidInput {
    explicit WrongType(std::string s) : InvalidInput(s) { reason = "Wrong type for parsing: "; throwMsg = reason + src; }
};
struct InvalidSeg : InvalidInput {
    explicit InvalidSeg(std::string s) : InvalidInput(s) {}
};
struct InvalidInputFile : InvalidSeg {
    explicit InvalidInputFile(std::string s) : InvalidSeg(s) { reason = "Invalid input file: "; throwMsg = reason + src; }
};
struct WrongNumberOfEntry : InvalidSeg {
    explicit WrongNumberOfEntry(std::string s) : InvalidSeg(s) { reason = "Number of variant site is wrong: "; throwMsg = reason + src; }
};
struct InvalidSegmentStartPosition : InvalidSeg {
    InvalidSegmentStartPosition(std::string a, std::string b) : InvalidSeg(a) {
        reason = "Segment start position at:";
        throwMsg = reason + src + std::string(" expect ") + b;
    }
};
struct NoDataError : InvalidSeg {
    NoDataError(std::string s, long long a, long long b) : InvalidSeg(s) {
        reason = "No data found in file ";
        throwMsg = reason + s + " between positions " + std::to_string(a) + " and " + std::to_string(b);
    }
};
// pattern.hpp:32-67
struct PatternDigitsExpected : InvalidInput {
    explicit PatternDigitsExpected(std::string s) : InvalidInput(s) {
        reason = "While parsing expression: expected digit as first character in number; got ";
        throwMsg = reason + src;
    }
};
struct PatternTimesExpected : InvalidInput {
    explicit PatternTimesExpected(std::string s) : InvalidInput(s) {
        reason = "While parsing expression: expected '*' to separate factors; got ";
        throwMsg = reason + src;
    }
};
struct PatternAddsExpected : InvalidInput {
    explicit PatternAddsExpected(std::string s) : InvalidInput(s) {
        reason = "While parsing expression: expected '+' to separate factors; got ";
        throwMsg = reason + src;
    }
};
struct Unsupported : InvalidInput {
    explicit Unsupported(std::string s) : InvalidInput(s) { throwMsg = "Not supported by this build: " + s; }
};

std::vector<std::string> expand_pattern(const std::string& pattern, double tmax);

// ---- the model tables handed to the device (what scrm's Model holds after Param::parse) ----
struct HostModel {
    double N0 = 10000;               // default_pop_size (-N0)
    int nsam = 2;
    int npop = 1;
    double loci_length = 2e7;
    double mutation_rate = 0;        // per bp per generation
    double recombination_rate = 0;   // per bp per generation
    std::vector<double> change_times;            // generations
    std::vector<std::vector<double>> pop_sizes;  // [E][P]
    std::vector<std::vector<double>> mig_rates;  // [E][P*P] per generation
    std::vector<std::vector<double>> single_mig; // [E][P*P]
    std::vector<std::vector<double>> coal_counts; // [E][P]   -vb event counts (1e10 when not given)
    std::vector<std::vector<double>> mig_counts;  // [E][P*P]
    std::vector<int> sample_pops;
    std::vector<double> bias_heights, bias_strengths;
    bool vb = false;
    uint64_t seed = 0;
    bool seed_set = false;
    double window_length_seq = 0;
    void parse(const std::vector<std::string>& tokens);   // tokens after "nsam nloci"
    void finalize();
};

enum Segment_State { SEGMENT_INVARIANT, SEGMENT_MISSING, SEGMENT_INVARIANT_PARTIAL };   // segdata.hpp:84

// One row of the resident table: `bases` bases starting at file position `first`; rows cut from one over-long file
// row share a genotype row.
struct SegPiece {
    long long first, bases;
    Segment_State kind;
    uint32_t geno;               // row of the genotype table
};

// per-row arrays of Segment::set_lookahead in the layout of pf_lookahead (include/smcsmc_pf.h)
struct LookaheadArrays {
    int max_doubletons = 1;
    std::vector<double> first_singleton_distance, relative_mutation_rate, doubleton_dist, first_split_distance;
    std::vector<int8_t> is_singleton_unphased, doubleton_idx, split_alleles;
    std::vector<int32_t> n_doubletons, split_count;
};

class Segment {   // the .seg input (counterpart of segdata.hpp:86-177)
  public:
    Segment(std::string file_name, size_t nsam, double seqlen, double num_of_mut, long long data_start = 1,
            double max_segment_length = 1e99);
    bool empty_file() const { return empty_file_; }
    size_t rows() const { return pieces_.size(); }
    // arrays for pf_load_segments (coordinates relative to data_start) with the per-row recording limit
    void pack(const std::vector<double>& lags, std::vector<double>& start, std::vector<double>& length,
              std::vector<int8_t>& state, std::vector<int8_t>& alleles, std::vector<int32_t>& max_record_epoch) const;
    // arrays for pf_load_lookahead: what Segment::set_lookahead (segdata.cpp:225-410) yields at every row
    void pack_lookahead(LookaheadArrays& out) const;
  private:
    void read_file();
    uint32_t add_genotype(std::string_view field);
    bool no_data_in(const SegPiece& p) const;
    std::string file_name_;
    size_t nsam_;
    long long data_start_;
    double seqlen_, max_segment_length_;
    bool empty_file_ = false;
    int field_width_ = -1;                 // characters in the genotype column, fixed by the first row
    std::vector<SegPiece> pieces_;
    std::vector<int8_t> genotypes_;        // [genotype row][nsam]: -1 missing, 0, 1, 2 unphased heterozygote
};

int max_epoch_to_update(const std::vector<double>& lags, double distance_to_mutation);   // smcsmc.cpp:266-275

struct EpochRange { int first = 0, last = 0; };     // closed, 0-based (-xr / -xc)

// Options of the driver, the parsed model and the output files of one run (the role of the reference's PfParam,
// pfparam.hpp:225-446).  Flags are applied from one table (pfparam.cpp: driver_options()).
class PfParam {
  public:
    static const int RECORD_RECOMB = 1;        // bits of record_mask (pfparam.hpp:279-281)
    static const int RECORD_COALMIGR = 2;
    void parse(int argc, char* argv[]);
    bool help() const { return want_help; }
    bool version() const { return want_version; }
    void print_help();
    void print_version(std::ostream* out);
    void write_out_header();
    void write_out_row(size_t iteration, int epoch, double epoch_begin, double epoch_end, std::string event_type, int from_pop,
                       int to_pop, double opportunity, double count, double weight);
    void write_resample_row(double position, double ess) const;
    int log();
    void write_log_text(std::ostream* out);
    void read_guide_file(const std::string& filename);

    // ---- options (defaults are the reference's, pfparam.cpp:193-255)
    size_t particles = 100;
    int em_iterations = 0;
    double ess_fraction = 0.5;
    bool ess_is_default = true;
    double lag = 0;
    bool calibrate_lag = true;
    double lag_fraction = 2.0;
    double delay = 0.5;
    int delay_type = 0;          // 0 recombination height (default), 1 coalescence, 2 coalescence or migration
    bool ancestral_aware = false, dephase = false;
    int apf_level = 0;
    double start_position = 1;
    double tmax = 2;
    bool cap_sizes = false;
    double size_cap = 200000;
    bool write_log_file = true, write_resample = false, record_trees = false;
    bool dump_model = false, dump_lookahead = false, dump_segments = false;
    bool want_help = false, want_version = false;
    size_t nsam = 2;
    double nodata_theta = 0;
    double row_cap_factor = 2.0;     // rows longer than this many expected recombination distances are cut
    std::string out_prefix = "smcsmc", seg_path, guide_path, pattern;
    std::vector<EpochRange> exclude_recomb, exclude_coalmigr;
    int chunks = 1, ranks = 0, devices = 0;          // several chunks in one process (main.cpp: run_chunks)
    bool delay_all = false;                          // -delay_all: pf_model.delay_type bit 2
    bool record_all = false;                         // -record_all: no recording limit far from data
    int mig_cap = 0;                                 // -migcap: pf_params.mig_cap (0 = the library's default)
    int delay_cap = 0;                               // -delaycap: pf_params.delay_cap (0 = the library's default)
    long long log_cap = 0;                           // -log_cap: pf_params.log_cap (0 = the library's default; -arg sizes its own)
    int count_wgs = 0;                               // -count_wgs: pf_params.count_wgs (0 = chosen here: the library's default, or tapered columns with six or more chunks per device)
    bool delay_evict = false;                        // -delay_evict: pf_params.flags bit 2
    std::string reduce_transport;                    // "rccl" / "host" / "" = choose
    double segment_cap() const;                      // rows longer than this many bases are cut (pfparam.cpp:364)
    // ---- derived
    std::string out_path, log_path, recomb_map_path, resample_path, trees_path;
    std::vector<int> record_mask;    // per epoch
    std::vector<std::string> model_tokens;
    HostModel model;
    Segment* segments = nullptr;
    size_t em_iteration = 0;
    // result of the M-step of the last E-step (CountModel::reset_model_parameters)
    std::vector<std::vector<double>> next_sizes, next_mig;
    double next_rho = 0;
    std::string cmdline;
    // recombination guide (RecombinationBias, pfparam.hpp:152-223): segment starts, sampling rates, relative leaf rates
    std::vector<double> guide_positions, guide_rates, guide_leaf_rates;

  private:
    void finalize();
};

std::string format_double(double d, double scientific_bound = 0.1, int precision = 2);   // pfparam.cpp:482-497
