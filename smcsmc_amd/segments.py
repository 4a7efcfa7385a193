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
