"""Host-side .seg handling, mirroring the reference's Segment class
(/root/reference/src/segdata.cpp, segdata.hpp) -- same names, same error behaviour.

  read_seg / prepare     Segment::prepare            segdata.cpp:55-166
  iterate rows           Segment::read_new_line      segdata.cpp:182-222
  distance_to_mutation   Segment::set_lookahead      segdata.cpp:234-262 (the part used without -apf)
  max_epoch_to_update    smcsmc.cpp:266-275
"""
import math

import numpy as np

SEGMENT_INVARIANT, SEGMENT_MISSING, SEGMENT_INVARIANT_PARTIAL = 0, 1, 2   # segdata.hpp:84


class InvalidSeg(ValueError):            # segdata.hpp:41-45
    pass


class InvalidInputFile(InvalidSeg):      # segdata.hpp:48-54
    def __init__(self, s):
        super().__init__("Invalid input file: " + s)


class WrongNumberOfEntry(InvalidSeg):    # segdata.hpp:57-63
    def __init__(self, s):
        super().__init__("Number of variant site is wrong: " + s)


class InvalidSegmentStartPosition(InvalidSeg):   # segdata.hpp:66-72
    def __init__(self, line, pos):
        super().__init__("Segment start position at:" + line + " expect " + pos)


class NoDataError(InvalidSeg):           # segdata.hpp:75-81
    def __init__(self, name, start, end):
        super().__init__("No data found in file %s between positions %d and %d" % (name, start, end))


_CODE = {".": -1, "/": 2, "0": 0, "1": 1}


def _strtol(s):
    """C strtol prefix parse: returns (value, rest)."""
    i = 0
    while i < len(s) and s[i] in " \t":
        i += 1
    j = i
    if j < len(s) and s[j] in "+-":
        j += 1
    k = j
    while k < len(s) and s[k].isdigit():
        k += 1
    if k == j:
        return 0, s
    return int(s[i:k]), s[k:]


class Segments:
    """The buffered SegDatum list plus the per-row quantities the filter consumes."""

    def __init__(self, file_name, nsam, seqlen, data_start=1, max_segment_length=1e99, num_of_mut=None):
        self.file_name = file_name
        self.nsam = int(nsam)
        self.seqlen = float(seqlen)
        self.data_start = int(data_start)
        self.max_segment_length = max_segment_length
        self.rows = []        # (segment_start, segment_length, state, alleles) in file coordinates
        self.empty_file = not file_name
        self._nfields = None
        if self.empty_file:
            # no-data mode (segdata.cpp:36-43, 175-178, 454-461)
            h = sum(1.0 / i for i in range(1, self.nsam))
            self.num_of_expected_mutations = h * num_of_mut
            seglen = math.ceil(int(self.seqlen) / self.num_of_expected_mutations)
            pos = 0
            while pos < self.seqlen:
                self.rows.append((pos + self.data_start, seglen, SEGMENT_MISSING, [-1] * self.nsam))
                pos += seglen
        else:
            self._prepare()

    # segdata.cpp:413-451
    def _extract_field_variant(self, field):
        if self.nsam > len(field):
            raise WrongNumberOfEntry(field)
        if self._nfields is None:
            self._nfields = len(field)
        elif self._nfields != len(field):
            raise WrongNumberOfEntry(field)
        out = []
        for i in range(self.nsam):
            ch = field[i]
            if ch not in _CODE:
                raise InvalidSeg("Unknown character found in .seg file; expect one of '.', '/', '0' or '1'.")
            out.append(_CODE[ch])
            if out[i] == -1 and i % 2 == 1 and out[i - 1] != -1:
                raise InvalidSeg("Found inconsistent unphased heterozygous marks")
        return out

    # segdata.cpp:55-166
    def _prepare(self):
        try:
            f = open(self.file_name, "r")
        except OSError:
            raise InvalidInputFile(self.file_name)
        next_start_pos = -1
        with f:
            for raw in f:
                line = raw.rstrip("\n")
                if len(line) == 0:
                    break                      # the first empty line ends the file
                if line[0] == "#":
                    continue
                cols = line.split("\t")
                if len(cols) < 3:
                    raise InvalidSeg("Require 3 or 6 columns")
                new_seg_start, rest = _strtol(line)
                if not rest.startswith("\t"):
                    raise InvalidSegmentStartPosition(line, str(new_seg_start))
                new_seg_len, _ = _strtol(cols[1])    # trailing ".0" tolerated (segdata.cpp:85-86)
                if cols[2] in ("T", "F"):
                    if len(cols) != 6:
                        raise InvalidSeg("Require 6 (or 3) columns")
                    if cols[3] not in ("T", "F"):
                        raise InvalidSeg("Expected T or F in .seg file column 3 and 4")
                    _, rest5 = _strtol(cols[4])
                    if rest5 != "" or cols[4] == "":
                        raise InvalidSeg("Bad chromosome (not an integer) in column 5")
                    allele = self._extract_field_variant(cols[5])
                else:
                    if len(cols) != 3:
                        raise InvalidSeg("Require 3 (or 6) columns")
                    allele = self._extract_field_variant(cols[2])
                if next_start_pos > -1 and next_start_pos != new_seg_start:
                    raise InvalidSeg("Segments are not consecutive")
                next_start_pos = new_seg_start + new_seg_len
                if new_seg_start >= self.data_start + self.seqlen:
                    break
                if new_seg_start + new_seg_len > self.data_start:
                    # split over-long segments (segdata.cpp:125-144)
                    while True:
                        if new_seg_len > self.max_segment_length:
                            new_seg_len = int(self.max_segment_length)
                            state = SEGMENT_INVARIANT_PARTIAL
                        else:
                            state = SEGMENT_INVARIANT
                        if new_seg_start + new_seg_len > self.data_start:
                            self.rows.append((new_seg_start, new_seg_len, state, allele))
                        new_seg_start += new_seg_len
                        new_seg_len = next_start_pos - new_seg_start
                        if not (new_seg_start < next_start_pos):
                            break
        if len(self.rows) == 0:
            raise NoDataError(self.file_name, self.data_start, int(self.data_start + self.seqlen))

    def __len__(self):
        return len(self.rows)

    # segdata.cpp:182-222 + 234-262 + smcsmc.cpp:266-275
    def pack(self, lags):
        """Arrays handed to the filter: coordinates relative to data_start (first base = 0)."""
        n = len(self.rows)
        start = np.zeros(n)
        length = np.zeros(n)
        state = np.zeros(n, np.int8)
        alleles = np.zeros((n, self.nsam), np.int8)
        cur = 0.0
        for i, (s, l, st, al) in enumerate(self.rows):
            ns = s - self.data_start
            ne = ns + l
            if ns < 0:
                ns = 0
            if ns > cur:
                raise InvalidSeg("Internal error - segment computation problem (start)")
            if ne < 0:
                raise InvalidSeg("Internal error - segment computation problem (end)")
            start[i] = ns
            length[i] = ne - ns
            state[i] = st
            alleles[i] = al
            cur = ne
        fstart = np.array([r[0] for r in self.rows], dtype=np.int64)
        flen = np.array([r[1] for r in self.rows], dtype=np.int64)
        dist = distance_to_mutation(fstart, flen, alleles)
        if self.empty_file:
            dist[:] = 0.0     # no-data mode never calls set_lookahead (segdata.cpp:189-191)
        mre = np.array([max_epoch_to_update(lags, d) for d in dist], np.int32)
        return {"start": start, "length": length, "state": state, "alleles": alleles,
                "max_record_epoch": mre, "distance_to_mutation": dist}


def distance_to_mutation(fstart, flen, alleles):
    """segdata.cpp:234-262: 0 for rows carrying data; inside an all-missing run the smaller of the
    distance back to the start of the run and forward to the end of the next row with data."""
    n = len(fstart)
    missing = (alleles == -1).all(axis=1)
    dist = np.zeros(n)
    # next row with data at or after i
    nxt = np.full(n, -1, np.int64)
    last = -1
    for i in range(n - 1, -1, -1):
        if not missing[i]:
            last = i
        nxt[i] = last
    run_start = 0
    for i in range(n):
        if not missing[i]:
            run_start = i + 1
            continue
        back = float(fstart[i] - fstart[run_start]) if run_start <= i else 0.0
        d = back
        if nxt[i] >= 0:
            fwd = float(fstart[nxt[i]] + flen[nxt[i]] - fstart[i])
            d = min(d, fwd)
        dist[i] = d
    return dist


def max_epoch_to_update(lags, distance):
    """smcsmc.cpp:266-275"""
    epoch = 0
    while epoch < len(lags) and distance < 0.5 * lags[epoch]:
        epoch += 1
    return epoch - 1


# ---------------------------------------------------------------------------------------------------------------
# Auxiliary particle filter look-ahead (Segment::set_lookahead, segdata.cpp:225-410)

MAX_MISSING_DATA = 2000000          # segdata.cpp:244
TBL_QUANTILES = (0.001, 0.003, 0.01, 0.03, 0.1, 0.5, 0.95)   # smcsmc.cpp:134


def set_lookahead(rows, cur, nsam):
    """Look-ahead summary for the row `cur` of the buffered SegDatum list (file coordinates), a line-by-line
    restatement of segdata.cpp:225-410: distance to the first singleton per lineage (negative: none seen within
    that distance), relative mutation rate under missing data, the doubletons (cherry evidence) with first and
    last evidence distances, and the first split."""
    fsd = [0.0] * nsam
    rmr = [0.0] * nsam
    doubleton = []            # [s1, s2, first_evidence, last_evidence, unphased_1, unphased_2, incompatible]
    first_split_distance = -1
    split_alleles = [0] * nsam
    split_count = 0
    found_doubleton = [False] * (nsam + 1)
    num_singletons = num_unphased_singletons = num_doubleton_sequences = 0
    tl = 0.1
    tl_missing = 0.1
    total_current_missing = 0.0
    last_singleton_distance = 0.0
    distance = 0.0
    unph = [False] * nsam
    start0 = rows[cur][0]
    for i in range(cur, len(rows)):
        seg_start, seg_len, _, al = rows[i]
        num_var = num_missing = 0
        s1 = s2 = -1
        unph = []
        j = 0
        while j < nsam:
            unph.append(False)
            if al[j] > 0:
                num_var += 1
                if num_var == 1:
                    s1 = j
                if num_var == 2:
                    s2 = j
                if al[j] == 2:
                    unph[j] = True
                    unph.append(True)
                    j += 1          # skip the second allele of an unphased het
            if j < nsam and al[j] == -1:
                num_missing += 1
                if num_missing == 1:
                    total_current_missing += seg_len
                if total_current_missing > MAX_MISSING_DATA:
                    if fsd[j] == 0:
                        eps = 1e-6
                        fsd[j] = -(seg_start - start0) - eps
                        last_singleton_distance = -fsd[j]
                        if fsd[j] < 0.5 * total_current_missing:
                            fsd[j] = -eps
                        rmr[j] = tl_missing / tl
                        num_singletons += 1
                    if not found_doubleton[j]:
                        found_doubleton[j] = True
                        num_doubleton_sequences += 1
            j += 1
        if num_missing == 0:
            total_current_missing = 0.0
        tl += seg_len * nsam
        tl_missing += seg_len * (nsam - num_missing)
        if total_current_missing > MAX_MISSING_DATA:
            continue
        have_doubleton = False
        distance = seg_start + seg_len - start0 + 0.5
        if num_var == 1:
            if fsd[s1] == 0:
                fsd[s1] = distance
                rmr[s1] = tl_missing / tl
                num_singletons += 1
                last_singleton_distance = fsd[s1]
                if unph[s1]:
                    fsd[s1 + 1] = distance
                    rmr[s1 + 1] = rmr[s1]
                    num_singletons += 1
                    num_unphased_singletons += 1
        else:
            for d in doubleton:
                a1, a2 = al[d[0]], al[d[1]]
                if ((d[0] | 1) == d[1] and a1 == 2) or ((a1 + a2 == 1) and ((a1 | a2) == 1)):
                    d[6] = True
                if num_var == 2 and d[0] == s1 and d[1] == s2:
                    have_doubleton = True
                    if not d[6]:
                        d[3] = distance
        if num_var == 2 and not have_doubleton and al[s1] > -1 and al[s2] > -1:
            done = False
            for d1 in range(0, (1 if al[s1] == 2 else 0) + 1):
                for d2 in range(0, (1 if al[s2] == 2 else 0) + 1):
                    if done:
                        continue
                    if not found_doubleton[s1 + d1] and not found_doubleton[s2 + d2]:
                        doubleton.append([s1, s2, distance, distance, al[s1] == 2, al[s2] == 2, False])
                        found_doubleton[s1 + d1] = True
                        num_doubleton_sequences += 1
                        found_doubleton[s2 + d2] = True
                        num_doubleton_sequences += 1
                        done = True
        if first_split_distance == -1 and num_var > 2 and nsam - num_var > 2:
            first_split_distance = distance
            split_alleles = list(al)
            split_count = min(num_var, nsam - num_var)
        if num_singletons == nsam and num_doubleton_sequences >= nsam - 1:
            break
        if num_singletons == nsam and distance > (2 + num_unphased_singletons) * last_singleton_distance:
            break
    if num_singletons < nsam:
        for j in range(nsam):
            if fsd[j] == 0:
                fsd[j] = -distance
                rmr[j] = tl_missing / tl
    unph = (list(unph) + [False] * nsam)[:nsam]
    return dict(first_singleton_distance=fsd, relative_mutation_rate=rmr, is_singleton_unphased=unph,
                doubleton=doubleton, first_split_distance=first_split_distance, split_alleles=split_alleles,
                split_count=split_count)


def pack_lookahead(rows, nsam):
    """Per-row look-ahead arrays in the layout of pf_lookahead (include/smcsmc_pf.h)."""
    S = len(rows)
    D = max(1, nsam // 2)
    out = dict(first_singleton_distance=np.zeros((S, nsam)), relative_mutation_rate=np.zeros((S, nsam)),
               is_singleton_unphased=np.zeros((S, nsam), np.int8), n_doubletons=np.zeros(S, np.int32),
               doubleton_idx=np.zeros((S, D, 4), np.int8), doubleton_dist=np.zeros((S, D, 2)),
               first_split_distance=np.full(S, -1.0), split_alleles=np.zeros((S, nsam), np.int8),
               split_count=np.zeros(S, np.int32), max_doubletons=D)
    for i in range(S):
        la = set_lookahead(rows, i, nsam)
        out["first_singleton_distance"][i] = la["first_singleton_distance"]
        out["relative_mutation_rate"][i] = la["relative_mutation_rate"]
        out["is_singleton_unphased"][i] = la["is_singleton_unphased"]
        nd = len(la["doubleton"])
        assert nd <= D
        out["n_doubletons"][i] = nd
        for k, d in enumerate(la["doubleton"]):
            out["doubleton_idx"][i, k] = [d[0], d[1], int(d[4]), int(d[5])]
            out["doubleton_dist"][i, k] = [d[2], d[3]]
        out["first_split_distance"][i] = la["first_split_distance"]
        out["split_alleles"][i] = la["split_alleles"]
        out["split_count"][i] = la["split_count"]
    return out


# ---------------------------------------------------------------------------------------------------------------
# Recombination guide files (RecombinationBias::parse_recomb_bias_file, pfparam.hpp:171-198)

def read_guide(path, nsam):
    """Reads a recombination guide (`locus size recomb_rate 1 .. n`, tab separated, plain or .gz, 0-based, no gaps)
    into the dict the model takes: positions[K], rates[K], leaf_rates[K][nsam]."""
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    try:
        f = opener(path, "rt")
    except OSError:
        raise InvalidSeg("Recombination guide file could not be opened.")
    pos, rates, leaves = [], [], []
    with f:
        header = f.readline()
        if header[:5] != "locus":
            raise InvalidSeg("Expected header line (with columns 'locus', 'size', 'recomb_rate', '1', ...) in recombination guide file")
        end = 0
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            if " " in line:
                raise InvalidSeg("Found spaces in recombination record")
            elts = line.split("\t")
            try:
                locus, size, rate = int(elts[0]), int(elts[1]), float(elts[2])
                lr = [float(v) for v in elts[3:]]
            except (ValueError, IndexError):
                raise InvalidSeg("Problem reading or parsing recombination guide file")
            if len(lr) != nsam:
                raise InvalidSeg("Did not find expected number of leaf columns")
            if locus != end:
                raise InvalidSeg("Did not get expected locus position (records should start at 0, and leave no gaps)")
            end = locus + size
            pos.append(float(locus)); rates.append(rate); leaves.append(lr)
    return dict(positions=np.array(pos), rates=np.array(rates), leaf_rates=np.array(leaves).reshape(len(pos), nsam))
