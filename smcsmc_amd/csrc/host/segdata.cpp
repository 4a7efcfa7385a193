// smcsmc_amd/csrc/host/segdata.cpp -- the .seg input of the drop-in binary: text file -> resident row table ->
// the flat arrays of pf_load_segments / pf_load_lookahead (include/smcsmc_pf.h).
//
// What is contract here is the file format and the per-row quantities the filter consumes (the
// reference's reader is /root/reference/src/segdata.cpp, its per-row look-ahead segdata.cpp:225-410 and the
// recording limit smcsmc.cpp:266-275).  How they are computed is this file's own:
//   * the file is read into memory once and rows are cut out of it as string_views; a row is a small struct, the
//     genotype strings are decoded into one flat int8 table that the pieces of a split row share;
//   * each genotype row is classified once (carriers, missing samples, unphased marks); the look-ahead of a row then
//     scans these classes instead of re-reading genotypes per (row, later row) pair;
//   * the distance from an all-missing row to the nearest data (which limits event recording) comes from one forward
//     and one backward sweep.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string_view>

#include "smcsmc_host.hpp"

namespace {

constexpr int kMaxFields = 8;

std::string slurp(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in.good()) throw InvalidInputFile(path);
    std::ostringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

// Cuts the next line (without its newline) off the front of `rest`.
std::string_view take_line(std::string_view& rest) {
    const size_t nl = rest.find('\n');
    std::string_view line = rest.substr(0, nl);
    rest.remove_prefix(nl == std::string_view::npos ? rest.size() : nl + 1);
    return line;
}

// Tab-separated fields of a line; a tab at the very end of the line does not open another field.
int split_fields(std::string_view line, std::string_view (&field)[kMaxFields]) {
    int count = 0;
    while (!line.empty()) {
        const size_t tab = line.find('\t');
        if (count < kMaxFields) field[count] = line.substr(0, tab);
        ++count;
        if (tab == std::string_view::npos) break;
        line.remove_prefix(tab + 1);
    }
    return count;
}

// Leading integer of a field ("521", "521.0" and "521 " all give 521); `whole` tells whether that was all of it.
long long leading_integer(std::string_view f, bool* whole) {
    long long v = 0;
    const char* b = f.data();
    const char* e = f.data() + f.size();
    while (b < e && (*b == ' ' || *b == '+')) ++b;
    auto r = std::from_chars(b, e, v);
    if (r.ec != std::errc()) { v = 0; r.ptr = b; }
    if (whole) *whole = (r.ptr == e);
    return v;
}

bool is_flag(std::string_view f) { return f == "T" || f == "F"; }

}  // namespace

// ------------------------------------------------------------------------------------------------ reading
Segment::Segment(std::string file_name, size_t nsam, double seqlen, double num_of_mut, long long data_start,
                 double max_segment_length)
    : file_name_(std::move(file_name)), nsam_(nsam), data_start_(data_start), seqlen_(seqlen),
      max_segment_length_(max_segment_length) {
    if (!file_name_.empty()) {
        read_file();
        return;
    }
    // No -seg: the filter runs over pseudo rows without data whose length is the expected distance between
    // segregating sites, ceil(L / (theta * H(n-1))).
    empty_file_ = true;
    double harmonic = 0.0;
    for (size_t i = 1; i < nsam_; ++i) harmonic += 1.0 / (double)i;
    long long step = (long long)std::ceil((double)(size_t)seqlen_ / (harmonic * num_of_mut));
    if (step < 1) step = 1;
    genotypes_.assign(nsam_, -1);
    for (long long at = 0; at < (long long)seqlen_; at += step)
        pieces_.push_back(SegPiece{at + data_start_, step, SEGMENT_MISSING, 0});
}

// Decodes one genotype field into a new row of genotypes_; returns its index.
uint32_t Segment::add_genotype(std::string_view field) {
    if (field.size() < nsam_) throw WrongNumberOfEntry(std::string(field));
    if (field_width_ < 0) {
        field_width_ = (int)field.size();
        if (field.size() != nsam_)
            std::cout << "Warning: analyzing " << nsam_ << " haplotypes, but input .seg file contains " << field.size()
                      << " haplotypes.  Ignoring remainder." << std::endl;
    } else if ((size_t)field_width_ != field.size()) {
        throw WrongNumberOfEntry(std::string(field));
    }
    const uint32_t row = (uint32_t)(genotypes_.size() / nsam_);
    for (size_t k = 0; k < nsam_; ++k) {
        int8_t code;
        switch (field[k]) {
            case '0': code = 0; break;
            case '1': code = 1; break;
            case '/': code = 2; break;      // unphased heterozygote
            case '.': code = -1; break;     // missing
            default: throw InvalidSeg("Unknown character found in .seg file; expect one of '.', '/', '0' or '1'.");
        }
        // the second haplotype of an individual cannot be missing when the first is not
        if (code == -1 && (k & 1) && genotypes_.back() != -1)
            throw InvalidSeg("Found inconsistent unphased heterozygous marks");
        genotypes_.push_back(code);
    }
    return row;
}

void Segment::read_file() {
    const std::string text = slurp(file_name_);
    const long long window_first = data_start_;
    const double window_end = (double)data_start_ + seqlen_;
    const long long piece_cap = max_segment_length_ >= 9e18 ? (long long)9e18 : (long long)max_segment_length_;
    std::string_view rest(text);
    long long expected_first = -1;          // first base the next row must start at
    std::string_view field[kMaxFields];
    while (!rest.empty()) {
        const std::string_view line = take_line(rest);
        if (line.empty()) break;            // an empty line ends the data
        if (line.front() == '#') continue;
        const int nf = split_fields(line, field);
        if (nf < 3) throw InvalidSeg("Require 3 or 6 columns");
        bool whole = false;
        const long long first = leading_integer(field[0], &whole);
        if (!whole) throw InvalidSegmentStartPosition(std::string(line), std::to_string(first));
        const long long bases = leading_integer(field[1], nullptr);
        std::string_view genotype;
        if (is_flag(field[2])) {
            // six columns: start, length, two flags, chromosome, genotypes
            if (nf != 6) throw InvalidSeg("Require 6 (or 3) columns");
            if (!is_flag(field[3])) throw InvalidSeg("Expected T or F in .seg file column 3 and 4");
            bool chrom_ok = true;
            if (!field[4].empty()) leading_integer(field[4], &chrom_ok);
            if (!chrom_ok) throw InvalidSeg("Bad chromosome (not an integer) in column 5");
            genotype = field[5];
        } else {
            if (nf != 3) throw InvalidSeg("Require 3 (or 6) columns");
            genotype = field[2];
        }
        const uint32_t geno = add_genotype(genotype);
        if (expected_first >= 0 && first != expected_first) throw InvalidSeg("Segments are not consecutive");
        expected_first = first + bases;
        if ((double)first >= window_end) break;
        // Rows longer than the cap are cut into capped pieces that carry no site (INVARIANT_PARTIAL) followed by the
        // remainder with the site at its end; pieces that end before the window are dropped.
        // A row of zero bases (two sites at one position) still carries its site: it is one empty piece.
        long long at = first;
        do {
            const long long left = expected_first - at;
            const bool capped = left > piece_cap;
            const long long take = capped ? piece_cap : left;
            if (at + take > window_first)
                pieces_.push_back(SegPiece{at, take, capped ? SEGMENT_INVARIANT_PARTIAL : SEGMENT_INVARIANT, geno});
            at += take;
        } while (at < expected_first);
    }
    if (pieces_.empty()) throw NoDataError(file_name_, data_start_, (long long)window_end);
}

// ------------------------------------------------------------------------------------------------ device arrays
int max_epoch_to_update(const std::vector<double>& lags, double distance_to_mutation) {
    // events of epoch e are worth recording only while data is nearer than half that epoch's lag; lags decrease with
    // the epoch, so the answer is the last epoch of the leading run that still qualifies (-1: none)
    int last = -1;
    for (size_t e = 0; e < lags.size() && distance_to_mutation < 0.5 * lags[e]; ++e) last = (int)e;
    return last;
}

bool Segment::no_data_in(const SegPiece& p) const {
    const int8_t* g = &genotypes_[(size_t)p.geno * nsam_];
    return std::all_of(g, g + nsam_, [](int8_t v) { return v == -1; });
}

void Segment::pack(const std::vector<double>& lags, std::vector<double>& start, std::vector<double>& length,
                   std::vector<int8_t>& state, std::vector<int8_t>& alleles, std::vector<int32_t>& max_record_epoch) const {
    const size_t rows = pieces_.size();
    start.resize(rows); length.resize(rows); state.resize(rows); alleles.resize(rows * nsam_); max_record_epoch.resize(rows);
    // coordinates relative to the first base of the window; the first piece may begin before it
    for (size_t i = 0; i < rows; ++i) {
        const SegPiece& p = pieces_[i];
        const long long lo = std::max<long long>(0, p.first - data_start_);
        const long long hi = p.first + p.bases - data_start_;
        start[i] = (double)lo;
        length[i] = (double)(hi - lo);
        state[i] = (int8_t)p.kind;
        std::copy_n(&genotypes_[(size_t)p.geno * nsam_], nsam_, &alleles[i * nsam_]);
    }
    // Distance from each row without data to the nearest data: to the left, back to the first row of its run of
    // data-free rows; to the right, to the end of the next row that has data.  Rows with data are at distance 0.
    std::vector<double> gap(rows, 0.0);
    long long run_first = -1;
    for (size_t i = 0; i < rows; ++i) {
        if (!no_data_in(pieces_[i])) { run_first = -1; continue; }
        if (run_first < 0) run_first = (long long)i;
        gap[i] = (double)(pieces_[i].first - pieces_[(size_t)run_first].first);
    }
    long long data_end = -1;                 // end of the nearest row with data to the right
    for (size_t k = rows; k-- > 0;) {
        if (!no_data_in(pieces_[k])) { data_end = pieces_[k].first + pieces_[k].bases; continue; }
        if (data_end >= 0) gap[k] = std::min(gap[k], (double)(data_end - pieces_[k].first));
    }
    for (size_t i = 0; i < rows; ++i) max_record_epoch[i] = max_epoch_to_update(lags, empty_file_ ? 0.0 : gap[i]);
}

// ------------------------------------------------------------------------------------------------ look-ahead
// What the auxiliary particle filter wants to know at a row (ForestState::includeLookaheadLikelihood,
// particle.cpp:439-617): per sample, how far ahead the next mutation carried by that sample alone lies (and what share
// of the bases up to there had data); the pairs of samples that next share a mutation nobody else carries, with the
// first and last evidence for each pair; the first mutation that splits the samples into two groups of three or more.
namespace {

// Everything the scan needs to know about one genotype row, derived once.
struct SiteClass {
    int carriers = 0;          // samples carrying the derived allele (an unphased pair counts once)
    int absent = 0;            // samples without data
    int first = -1, second = -1;   // the first two carriers
    uint64_t unphased = 0;     // bit j: sample j opened an unphased pair at this row
    uint64_t no_data = 0;      // bit j: sample j is missing (as seen by the scan, see classify())
    uint64_t flags_len = 0;    // how many unphased flags the row produces (nsam + one per unphased pair)
};

SiteClass classify(const int8_t* g, int nsam) {
    SiteClass c;
    int j = 0;
    while (j < nsam) {
        if (g[j] > 0) {
            ++c.carriers;
            if (c.carriers == 1) c.first = j;
            if (c.carriers == 2) c.second = j;
            if (g[j] == 2) {
                // an unphased heterozygote: its partner haplotype is not looked at as a carrier
                c.unphased |= 1ull << j;
                ++j;
            }
        }
        // the missing-data test follows the carrier test and therefore sees the partner of an unphased pair
        if (j < nsam && g[j] == -1) { ++c.absent; c.no_data |= 1ull << j; }
        ++j;
    }
    return c;
}

struct SharedPair { int a, b; double first_seen, last_seen; bool a_unphased, b_unphased, refuted; };

// State of one forward scan.
struct Ahead {
    std::vector<double> next_private, data_share;
    std::vector<SharedPair> pairs;
    std::vector<char> in_pair;
    double split_at = -1.0;
    uint32_t split_row = 0;
    int split_minor = 0;
    int have_private = 0, have_private_unphased = 0, samples_in_pairs = 0;
    double bases_seen = 0.1, bases_with_data = 0.1;     // x samples; the offset keeps the ratio defined
    double missing_streak = 0.0, last_private = 0.0, reach = 0.0;
    explicit Ahead(int nsam) : next_private(nsam, 0.0), data_share(nsam, 0.0), in_pair(nsam + 1, 0) {}
    double share() const { return bases_with_data / bases_seen; }
};

constexpr double kGiveUpMissing = 2000000.0;   // a sample without data for this long is not waited for
constexpr double kTiny = 1e-6;

}  // namespace

void Segment::pack_lookahead(LookaheadArrays& out) const {
    const size_t rows = pieces_.size();
    const int nsam = (int)nsam_;
    const int slots = std::max(1, nsam / 2);
    out.max_doubletons = slots;
    out.first_singleton_distance.assign(rows * nsam, 0.0);
    out.relative_mutation_rate.assign(rows * nsam, 0.0);
    out.is_singleton_unphased.assign(rows * nsam, 0);
    out.n_doubletons.assign(rows, 0);
    out.doubleton_idx.assign(rows * slots * 4, 0);
    out.doubleton_dist.assign(rows * slots * 2, 0.0);
    out.first_split_distance.assign(rows, -1.0);
    out.split_alleles.assign(rows * nsam, 0);
    out.split_count.assign(rows, 0);

    const size_t ngeno = genotypes_.size() / nsam_;
    std::vector<SiteClass> cls(ngeno);
    for (size_t g = 0; g < ngeno; ++g) cls[g] = classify(&genotypes_[g * nsam_], nsam);

    for (size_t here = 0; here < rows; ++here) {
        Ahead st(nsam);
        const long long origin = pieces_[here].first;
        uint32_t last_looked_at = pieces_[here].geno;
        for (size_t i = here; i < rows; ++i) {
            const SegPiece& pc = pieces_[i];
            const SiteClass& c = cls[pc.geno];
            const int8_t* g = &genotypes_[(size_t)pc.geno * nsam_];
            last_looked_at = pc.geno;
            // samples without data: the streak of bases over which somebody was missing, and giving up on samples
            if (c.absent > 0) {
                st.missing_streak += (double)pc.bases;
                if (st.missing_streak > kGiveUpMissing)
                    for (int j = 0; j < nsam; ++j) {
                        if (!((c.no_data >> j) & 1)) continue;
                        if (st.next_private[j] == 0) {
                            // marked "none seen" by a negative distance; when most of the way was missing, by -tiny
                            double mark = -(double)(pc.first - origin) - kTiny;
                            st.last_private = -mark;
                            if (mark < 0.5 * st.missing_streak) mark = -kTiny;
                            st.next_private[j] = mark;
                            st.data_share[j] = st.share();
                            ++st.have_private;
                        }
                        if (!st.in_pair[j]) { st.in_pair[j] = 1; ++st.samples_in_pairs; }
                    }
            } else {
                st.missing_streak = 0.0;
            }
            st.bases_seen += (double)pc.bases * nsam;
            st.bases_with_data += (double)pc.bases * (nsam - c.absent);
            if (st.missing_streak > kGiveUpMissing) continue;
            st.reach = (double)(pc.first + pc.bases - origin) + 0.5;

            if (c.carriers == 1) {
                const int j = c.first;
                if (st.next_private[j] == 0) {
                    st.next_private[j] = st.reach;
                    st.data_share[j] = st.share();
                    st.last_private = st.reach;
                    ++st.have_private;
                    if ((c.unphased >> j) & 1) {      // either haplotype of the individual may carry it
                        st.next_private[j + 1] = st.reach;
                        st.data_share[j + 1] = st.data_share[j];
                        ++st.have_private;
                        ++st.have_private_unphased;
                    }
                }
            } else {
                bool known = false;
                for (SharedPair& p : st.pairs) {
                    const int ga = g[p.a], gb = g[p.b];
                    const bool same_individual_het = ((p.a | 1) == p.b) && ga == 2;
                    const bool differ = (ga + gb == 1) && ((ga | gb) == 1);
                    if (same_individual_het || differ) p.refuted = true;
                    if (c.carriers == 2 && p.a == c.first && p.b == c.second) {
                        known = true;
                        if (!p.refuted) p.last_seen = st.reach;
                    }
                }
                if (c.carriers == 2 && !known && g[c.first] > -1 && g[c.second] > -1) {
                    // a new shared mutation; with unphased carriers either haplotype may be the free one
                    const int ua = g[c.first] == 2, ub = g[c.second] == 2;
                    bool placed = false;
                    for (int da = 0; da <= ua && !placed; ++da)
                        for (int db = 0; db <= ub && !placed; ++db)
                            if (!st.in_pair[c.first + da] && !st.in_pair[c.second + db]) {
                                st.pairs.push_back(SharedPair{c.first, c.second, st.reach, st.reach, ua != 0, ub != 0, false});
                                st.in_pair[c.first + da] = 1;
                                st.in_pair[c.second + db] = 1;
                                st.samples_in_pairs += 2;
                                placed = true;
                            }
                }
            }
            if (st.split_at == -1.0 && c.carriers > 2 && nsam - c.carriers > 2) {
                st.split_at = st.reach;
                st.split_row = pc.geno;
                st.split_minor = std::min(c.carriers, nsam - c.carriers);
            }
            if (st.have_private == nsam) {
                if (st.samples_in_pairs >= nsam - 1) break;
                if (st.reach > (2 + st.have_private_unphased) * st.last_private) break;     // far beyond the last one
            }
        }
        // ran off the data before every sample had its own mutation: "none within <reach>"
        if (st.have_private < nsam)
            for (int j = 0; j < nsam; ++j)
                if (st.next_private[j] == 0) { st.next_private[j] = -st.reach; st.data_share[j] = st.share(); }

        const SiteClass& lastc = cls[last_looked_at];
        for (int j = 0; j < nsam; ++j) {
            out.first_singleton_distance[here * nsam + j] = st.next_private[j];
            out.relative_mutation_rate[here * nsam + j] = st.data_share[j];
            out.is_singleton_unphased[here * nsam + j] = (int8_t)((lastc.unphased >> j) & 1);
            if (st.split_at != -1.0) out.split_alleles[here * nsam + j] = genotypes_[(size_t)st.split_row * nsam_ + j];
        }
        if ((int)st.pairs.size() > slots) throw InvalidSeg("Internal error - more doubletons than sequence pairs");
        out.n_doubletons[here] = (int32_t)st.pairs.size();
        for (size_t k = 0; k < st.pairs.size(); ++k) {
            const SharedPair& p = st.pairs[k];
            int8_t* idx = &out.doubleton_idx[(here * slots + k) * 4];
            idx[0] = (int8_t)p.a; idx[1] = (int8_t)p.b; idx[2] = p.a_unphased; idx[3] = p.b_unphased;
            out.doubleton_dist[(here * slots + k) * 2] = p.first_seen;
            out.doubleton_dist[(here * slots + k) * 2 + 1] = p.last_seen;
        }
        out.first_split_distance[here] = st.split_at;
        out.split_count[here] = st.split_minor;
    }
}
