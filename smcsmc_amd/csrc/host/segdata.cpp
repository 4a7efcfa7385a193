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
