// smcsmc_amd/csrc/host/segdata.cpp -- .seg reader; follows /root/reference/src/segdata.cpp line by line
// in behaviour (prepare 55-166, extract_field_VARIANT 413-451, read_new_line 182-222, the
// distance_to_mutation part of set_lookahead 234-262) and smcsmc.cpp:266-275.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "smcsmc_host.hpp"

using namespace std;

Segment::Segment(string file_name, size_t nsam, double seqlen, double num_of_mut, long long data_start, double max_segment_length)
    : file_name_(file_name), nsam_(nsam), data_start_(data_start), seqlen_(seqlen), max_segment_length_(max_segment_length) {
    if (file_name_.size() == 0) {
        // no-data mode: all-missing pseudo segments (segdata.cpp:36-43, 175-178, 454-461)
        empty_file_ = true;
        double s = 0.0;
        for (size_t i = 1; i < nsam; i++) s += 1.0 / i;
        num_of_expected_mutations_ = s * num_of_mut;
        long long seglen = (long long)ceil((size_t)seqlen_ / num_of_expected_mutations_);
        if (seglen < 1) seglen = 1;
        for (long long pos = 0; pos < (long long)seqlen_; pos += seglen)
            buffer_.push_back(SegDatum{pos + data_start_, seglen, SEGMENT_MISSING, vector<int>(nsam_, -1)});
    } else {
        prepare();
    }
}

vector<int> Segment::extract_field_VARIANT(const string& field) {
    vector<int> out;
    if (nsam_ > field.size()) throw WrongNumberOfEntry(field);
    if (number_of_fields_ == -1) {
        number_of_fields_ = (int)field.size();
        if (nsam_ != field.size())
            cout << "Warning: analyzing " << nsam_ << " haplotypes, but input .seg file contains " << field.size()
                 << " haplotypes.  Ignoring remainder." << endl;
    } else if (number_of_fields_ != (int)field.size()) {
        throw WrongNumberOfEntry(field);
    }
    for (size_t i = 0; i < nsam_; i++) {
        int v;
        switch (field[i]) {
            case '.': v = -1; break;
            case '/': v = 2; break;
            case '0': v = 0; break;
            case '1': v = 1; break;
            default: throw InvalidSeg("Unknown character found in .seg file; expect one of '.', '/', '0' or '1'.");
        }
        out.push_back(v);
        if (v == -1 && (i % 2) == 1 && out[i - 1] != -1) throw InvalidSeg("Found inconsistent unphased heterozygous marks");
    }
    return out;
}

void Segment::prepare() {
    ifstream in(file_name_.c_str());
    if (!in.good()) throw InvalidInputFile(file_name_);
    string line;
    long long next_start_pos = -1;
    getline(in, line);
    while (line.size() > 0) {
        if (line[0] != '#') {
            vector<int> col_starts;
            int pos = 0;
            while (line[pos] && line[pos] != '\n') {
                col_starts.push_back(pos);
                for (; line[pos] && line[pos] != '\n' && line[pos] != '\t'; ++pos) {}
                if (line[pos] == '\t') ++pos;
            }
            if (col_starts.size() < 3) throw InvalidSeg("Require 3 or 6 columns");
            char* end_ptr;
            vector<int> allele;
            long long new_seg_start = strtoll(line.c_str(), &end_ptr, 10);
            if (*end_ptr != '\t') throw InvalidSegmentStartPosition(line, to_string(new_seg_start));
            long new_seg_len = strtol(line.c_str() + col_starts[1], &end_ptr, 10);
            if ((line[col_starts[2]] == 'T' || line[col_starts[2]] == 'F') && line[col_starts[2] + 1] == '\t') {
                if (col_starts.size() != 6) throw InvalidSeg("Require 6 (or 3) columns");
                if ((line[col_starts[3]] != 'T' && line[col_starts[3]] != 'F') || line[col_starts[3] + 1] != '\t')
                    throw InvalidSeg("Expected T or F in .seg file column 3 and 4");
                strtol(line.c_str() + col_starts[4], &end_ptr, 10);
                if (*end_ptr != '\t') throw InvalidSeg("Bad chromosome (not an integer) in column 5");
                allele = extract_field_VARIANT(string(line.c_str() + col_starts[5]));
            } else {
                if (col_starts.size() != 3) throw InvalidSeg("Require 3 (or 6) columns");
                allele = extract_field_VARIANT(string(line.c_str() + col_starts[2]));
            }
            if (next_start_pos > -1 && next_start_pos != new_seg_start) throw InvalidSeg("Segments are not consecutive");
            next_start_pos = new_seg_start + new_seg_len;
            if (new_seg_start >= data_start_ + seqlen_) break;
            Segment_State state;
            if (new_seg_start + new_seg_len > data_start_) {
                do {
                    if (new_seg_len > max_segment_length_) {
                        new_seg_len = (long)max_segment_length_;
                        state = SEGMENT_INVARIANT_PARTIAL;
                    } else {
                        state = SEGMENT_INVARIANT;
                    }
                    if (new_seg_start + new_seg_len > data_start_)
                        buffer_.push_back(SegDatum{new_seg_start, new_seg_len, state, allele});
                    new_seg_start += new_seg_len;
                    new_seg_len = next_start_pos - new_seg_start;
                } while (new_seg_start < next_start_pos);
            }
        }
        line.clear();
        getline(in, line);
    }
    if (buffer_.size() == 0) throw NoDataError(file_name_, data_start_, (long long)(data_start_ + seqlen_));
}

int max_epoch_to_update(const vector<double>& lags, double distance_to_mutation) {
    int epoch = 0;
    const double scale_factor = 0.5;
    while (epoch < (int)lags.size() && distance_to_mutation < scale_factor * lags[epoch]) epoch++;
    return epoch - 1;
}

void Segment::pack(const vector<double>& lags, vector<double>& start, vector<double>& length, vector<int8_t>& state,
                   vector<int8_t>& alleles, vector<int32_t>& mre) const {
    const size_t n = buffer_.size();
    start.resize(n); length.resize(n); state.resize(n); alleles.resize(n * nsam_); mre.resize(n);
    double segment_start = 0, segment_length = 0;
    for (size_t i = 0; i < n; ++i) {
        const SegDatum& sd = buffer_[i];
        segment_start += segment_length;                       // read_new_line (segdata.cpp:187)
        long new_seg_start = (long)(sd.segment_start - data_start_);
        long new_seg_end = new_seg_start + sd.segment_length;
        if (new_seg_start < 0) new_seg_start = 0;
        if (new_seg_start > segment_start) throw InvalidSeg("Internal error - segment computation problem (start)");
        if (new_seg_end < 0) throw InvalidSeg("Internal error - segment computation problem (end)");
        segment_start = new_seg_start;
        segment_length = new_seg_end - new_seg_start;
        start[i] = segment_start; length[i] = segment_length; state[i] = (int8_t)sd.segment_state;
        for (size_t k = 0; k < nsam_; ++k) alleles[i * nsam_ + k] = (int8_t)sd.allele_state[k];
    }
    // distance_to_mutation (segdata.cpp:234-262) -> max_epoch_to_update
    vector<long long> next_data(n, -1);
    long long last = -1;
    for (long long i = (long long)n - 1; i >= 0; --i) {
        if (!buffer_[i].all_alleles_missing()) last = i;
        next_data[i] = last;
    }
    size_t run_start = 0;
    for (size_t i = 0; i < n; ++i) {
        double d = 0;
        if (!buffer_[i].all_alleles_missing()) {
            run_start = i + 1;
        } else {
            d = (double)(buffer_[i].segment_start - buffer_[run_start].segment_start);
            if (next_data[i] >= 0) {
                const SegDatum& nd = buffer_[next_data[i]];
                d = std::min(d, (double)(nd.segment_start + nd.segment_length - buffer_[i].segment_start));
            }
        }
        if (empty_file_) d = 0;   // no-data mode never calls set_lookahead (segdata.cpp:189-191)
        mre[i] = max_epoch_to_update(lags, d);
    }
}


// ------------------------------------------------------------------ auxiliary particle filter look-ahead
// Segment::set_lookahead (segdata.cpp:225-410) for every buffered row, written into the arrays of pf_lookahead.
void Segment::pack_lookahead(LookaheadArrays& out) const {
    const size_t S = buffer_.size();
    const int nsam = (int)nsam_;
    const int D = std::max(1, nsam / 2);
    out.max_doubletons = D;
    out.first_singleton_distance.assign(S * nsam, 0.0);
    out.relative_mutation_rate.assign(S * nsam, 0.0);
    out.is_singleton_unphased.assign(S * nsam, 0);
    out.n_doubletons.assign(S, 0);
    out.doubleton_idx.assign(S * D * 4, 0);
    out.doubleton_dist.assign(S * D * 2, 0.0);
    out.first_split_distance.assign(S, -1.0);
    out.split_alleles.assign(S * nsam, 0);
    out.split_count.assign(S, 0);
    struct Doubleton { int s1, s2; double first, last; bool u1, u2, incompatible; };
    const double max_missing_data = 2000000;
    for (size_t cur = 0; cur < S; ++cur) {
        vector<double> fsd(nsam, 0.0), rmr(nsam, 0.0);
        vector<Doubleton> doubleton;
        double first_split_distance = -1;
        vector<int> split_alleles(nsam, 0);
        int split_count = 0;
        vector<bool> found_doubleton(nsam + 1, false);
        int num_singletons = 0, num_unphased_singletons = 0, num_doubleton_sequences = 0;
        double tl = 0.1, tl_missing = 0.1, total_current_missing = 0.0, last_singleton_distance = 0.0, distance = 0.0;
        vector<int> unph;
        const long long start0 = buffer_[cur].segment_start;
        for (size_t i = cur; i < S; i++) {
            const SegDatum& b = buffer_[i];
            int num_var = 0, num_missing = 0, s1 = -1, s2 = -1;
            unph.clear();
            for (int j = 0; j < nsam; j++) {
                unph.push_back(0);
                if (b.allele_state[j] > 0) {
                    num_var++;
                    if (num_var == 1) s1 = j;
                    if (num_var == 2) s2 = j;
                    if (b.allele_state[j] == 2) {
                        unph[j] = 1;
                        unph.push_back(1);
                        j++;
                    }
                }
                if (j < nsam && b.allele_state[j] == -1) {
                    num_missing++;
                    if (num_missing == 1) total_current_missing += b.segment_length;
                    if (total_current_missing > max_missing_data) {
                        if (fsd[j] == 0) {
                            const double epsilon = 1e-6;
                            fsd[j] = -(double)(b.segment_start - start0) - epsilon;
                            last_singleton_distance = -fsd[j];
                            if (fsd[j] < 0.5 * total_current_missing) fsd[j] = -epsilon;
                            rmr[j] = tl_missing / tl;
                            num_singletons++;
                        }
                        if (!found_doubleton[j]) { found_doubleton[j] = true; num_doubleton_sequences++; }
                    }
                }
            }
            if (num_missing == 0) total_current_missing = 0.0;
            tl += (double)b.segment_length * nsam;
            tl_missing += (double)b.segment_length * (nsam - num_missing);
            if (total_current_missing > max_missing_data) continue;
            bool have_doubleton = false;
            distance = (double)(b.segment_start + b.segment_length - start0) + 0.5;
            if (num_var == 1) {
                if (fsd[s1] == 0) {
                    fsd[s1] = distance;
                    rmr[s1] = tl_missing / tl;
                    num_singletons++;
                    last_singleton_distance = fsd[s1];
                    if (unph[s1]) {
                        fsd[s1 + 1] = distance;
                        rmr[s1 + 1] = rmr[s1];
                        num_singletons++;
                        num_unphased_singletons++;
                    }
                }
            } else {
                for (Doubleton& d : doubleton) {
                    int a1 = b.allele_state[d.s1], a2 = b.allele_state[d.s2];
                    if (((d.s1 | 1) == d.s2 && a1 == 2) || ((a1 + a2 == 1) && ((a1 | a2) == 1))) d.incompatible = true;
                    if (num_var == 2 && d.s1 == s1 && d.s2 == s2) {
                        have_doubleton = true;
                        if (!d.incompatible) d.last = distance;
                    }
                }
            }
            if (num_var == 2 && !have_doubleton && b.allele_state[s1] > -1 && b.allele_state[s2] > -1) {
                for (int d1 = 0; d1 <= (b.allele_state[s1] == 2); d1++)
                    for (int d2 = 0; d2 <= (b.allele_state[s2] == 2); d2++)
                        if (!found_doubleton[s1 + d1] && !found_doubleton[s2 + d2]) {
                            doubleton.push_back(Doubleton{s1, s2, distance, distance, b.allele_state[s1] == 2,
                                                          b.allele_state[s2] == 2, false});
                            found_doubleton[s1 + d1] = true; num_doubleton_sequences++;
                            found_doubleton[s2 + d2] = true; num_doubleton_sequences++;
                            d1 = 1; d2 = 1;
                        }
            }
            if (first_split_distance == -1 && num_var > 2 && nsam - num_var > 2) {
                first_split_distance = distance;
                split_alleles = b.allele_state;
                split_count = std::min(num_var, nsam - num_var);
            }
            if ((num_singletons == nsam) && num_doubleton_sequences >= nsam - 1) break;
            if ((num_singletons == nsam) && distance > (2 + num_unphased_singletons) * last_singleton_distance) break;
        }
        if (num_singletons < nsam)
            for (int j = 0; j < nsam; j++)
                if (fsd[j] == 0) { fsd[j] = -distance; rmr[j] = tl_missing / tl; }
        unph.resize(nsam, 0);
        for (int j = 0; j < nsam; ++j) {
            out.first_singleton_distance[cur * nsam + j] = fsd[j];
            out.relative_mutation_rate[cur * nsam + j] = rmr[j];
            out.is_singleton_unphased[cur * nsam + j] = (int8_t)unph[j];
            out.split_alleles[cur * nsam + j] = (int8_t)split_alleles[j];
        }
        if ((int)doubleton.size() > D) throw InvalidSeg("Internal error - more doubletons than sequence pairs");
        out.n_doubletons[cur] = (int32_t)doubleton.size();
        for (size_t k = 0; k < doubleton.size(); ++k) {
            int8_t* di = &out.doubleton_idx[(cur * D + k) * 4];
            di[0] = (int8_t)doubleton[k].s1; di[1] = (int8_t)doubleton[k].s2;
            di[2] = doubleton[k].u1; di[3] = doubleton[k].u2;
            out.doubleton_dist[(cur * D + k) * 2] = doubleton[k].first;
            out.doubleton_dist[(cur * D + k) * 2 + 1] = doubleton[k].last;
        }
        out.first_split_distance[cur] = first_split_distance;
        out.split_count[cur] = split_count;
    }
}
