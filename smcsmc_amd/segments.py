"""Host-side .seg handling for the Python binding: the same row table and per-row quantities as the C++ reader of
the drop-in binary (smcsmc_amd/csrc/host/segdata.cpp), built on numpy arrays.

Contract (reference: src/segdata.cpp, smcsmc.cpp:266-275): the .seg text format with its error
messages, the splitting of over-long rows, the window selected by data_start / seqlen, the per-row recording limit
and the look-ahead summaries of the auxiliary particle filter.  The representation is this module's own: a row table
of four arrays (`first`, `bases`, `kind`, `genotype`), a per-row classification computed once, and a forward scan
over those classes for the look-ahead.
"""
import math

import numpy as np

SEGMENT_INVARIANT, SEGMENT_MISSING, SEGMENT_INVARIANT_PARTIAL = 0, 1, 2   # segdata.hpp:84


class InvalidSeg(ValueError):
    pass


class InvalidInputFile(InvalidSeg):
    def __init__(self, s):
        super().__init__("Invalid input file: " + s)


class WrongNumberOfEntry(InvalidSeg):
    def __init__(self, s):
        super().__init__("Number of variant site is wrong: " + s)


class InvalidSegmentStartPosition(InvalidSeg):
    def __init__(self, line, pos):
        super().__init__("Segment start position at:" + line + " expect " + pos)


class NoDataError(InvalidSeg):
    def __init__(self, name, start, end):
        super().__init__("No data found in file %s between positions %d and %d" % (name, start, end))


_GENOTYPE_CODE = np.full(256, 99, np.int8)
for _ch, _v in ((".", -1), ("0", 0), ("1", 1), ("/", 2)):
    _GENOTYPE_CODE[ord(_ch)] = _v


def _leading_int(text):
    """(value, whole): the integer a field starts with ("521.0" -> 521) and whether that was the whole field."""
    t = text.lstrip(" +")
    k = 1 if t[:1] == "-" else 0
    while k < len(t) and t[k].isdigit():
        k += 1
    digits = t[:k]
    if digits in ("", "-"):
        return 0, len(t) == 0            # an empty field reads as 0
    return int(digits), k == len(t)


def split_long_rows(first, bases, cap):
    """Cuts rows longer than `cap` bases into capped pieces followed by the remainder.  Returns (first, bases, kind,
    source row) of the pieces; capped pieces are INVARIANT_PARTIAL (no site at their end), a row of zero bases stays
    one empty piece."""
    first = np.asarray(first, np.int64)
    bases = np.asarray(bases, np.int64)
    cap = int(min(cap, 2 ** 62))
    pieces = np.maximum(1, -(-bases // cap))                 # ceil, at least one
    src = np.repeat(np.arange(len(first)), pieces)
    nth = np.arange(len(src)) - np.repeat(np.cumsum(pieces) - pieces, pieces)
    last = nth == pieces[src] - 1
    p_first = first[src] + nth * cap
    p_bases = np.where(last, bases[src] - nth * cap, cap)
    kind = np.where(last, SEGMENT_INVARIANT, SEGMENT_INVARIANT_PARTIAL).astype(np.int8)
    return p_first, p_bases, kind, src


class Segments:
    """The resident row table of one chunk."""

    def __init__(self, file_name, nsam, seqlen, data_start=1, max_segment_length=1e99, num_of_mut=None):
        self.file_name = file_name
        self.nsam = int(nsam)
        self.seqlen = float(seqlen)
        self.data_start = int(data_start)
        self.max_segment_length = max_segment_length
        self.empty_file = not file_name
        if self.empty_file:
            # no -seg: pseudo rows without data, ceil(L / (theta * H(n-1))) bases each
            harmonic = sum(1.0 / i for i in range(1, self.nsam))
            step = max(1, math.ceil(int(self.seqlen) / (harmonic * num_of_mut)))
            self.first = np.arange(0, int(math.ceil(self.seqlen)), step, dtype=np.int64) + self.data_start
            self.bases = np.full(len(self.first), step, np.int64)
            self.kind = np.full(len(self.first), SEGMENT_MISSING, np.int8)
            self.genotype = np.full((len(self.first), self.nsam), -1, np.int8)
        else:
            self._read(file_name)

    @classmethod
    def from_sites(cls, first, bases, genotype, nsam, seqlen, data_start=1, max_segment_length=1e99):
        """A table from in-memory site data (what a simulator produces): rows are split like rows read from a file."""
        self = cls.__new__(cls)
        self.file_name = "<memory>"
        self.nsam = int(nsam); self.seqlen = float(seqlen); self.data_start = int(data_start)
        self.max_segment_length = max_segment_length
        self.empty_file = False
        self._finish(np.asarray(first, np.int64), np.asarray(bases, np.int64),
                     np.asarray(genotype, np.int8).reshape(-1, self.nsam))
        return self

    @classmethod
    def from_pieces(cls, first, bases, kind, genotype, nsam, seqlen, data_start=1):
        """A table from rows that are already cut (e.g. a slice of another table)."""
        self = cls.__new__(cls)
        self.file_name = "<memory>"
        self.nsam = int(nsam); self.seqlen = float(seqlen); self.data_start = int(data_start)
        self.max_segment_length = 1e99
        self.empty_file = False
        self.first = np.asarray(first, np.int64); self.bases = np.asarray(bases, np.int64)
        self.kind = np.asarray(kind, np.int8); self.genotype = np.asarray(genotype, np.int8).reshape(-1, self.nsam)
        return self

    # ---- reading
    def _decode(self, field, width):
        if len(field) < self.nsam:
            raise WrongNumberOfEntry(field)
        if width[0] is None:
            width[0] = len(field)
        elif width[0] != len(field):
            raise WrongNumberOfEntry(field)
        g = _GENOTYPE_CODE[np.frombuffer(field[:self.nsam].encode("latin-1"), np.uint8)]
        if (g == 99).any():
            raise InvalidSeg("Unknown character found in .seg file; expect one of '.', '/', '0' or '1'.")
        # the second haplotype of an individual cannot be missing when the first is not
        second = g[1::2]
        if ((second == -1) & (g[0:2 * len(second):2] != -1)).any():
            raise InvalidSeg("Found inconsistent unphased heterozygous marks")
        return g

    def _read(self, path):
        try:
            f = open(path, "r")
        except OSError:
            raise InvalidInputFile(path)
        first, bases, geno = [], [], []
        width = [None]
        expected = None
        window_end = self.data_start + self.seqlen
        with f:
            for raw in f:
                line = raw.rstrip("\n")
                if not line:
                    break                                  # an empty line ends the data
                if line[0] == "#":
                    continue
                field = line.split("\t")
                if field[-1] == "" and len(field) > 1:
                    field.pop()                            # a tab at the very end opens no field
                if len(field) < 3:
                    raise InvalidSeg("Require 3 or 6 columns")
                at, whole = _leading_int(field[0])
                if not whole:
                    raise InvalidSegmentStartPosition(line, str(at))
                count, _ = _leading_int(field[1])
                if field[2] in ("T", "F"):
                    if len(field) != 6:
                        raise InvalidSeg("Require 6 (or 3) columns")
                    if field[3] not in ("T", "F"):
                        raise InvalidSeg("Expected T or F in .seg file column 3 and 4")
                    if field[4] != "" and not _leading_int(field[4])[1]:
                        raise InvalidSeg("Bad chromosome (not an integer) in column 5")
                    g = self._decode(field[5], width)
                else:
                    if len(field) != 3:
                        raise InvalidSeg("Require 3 (or 6) columns")
                    g = self._decode(field[2], width)
                if expected is not None and at != expected:
                    raise InvalidSeg("Segments are not consecutive")
                expected = at + count
                if at >= window_end:
                    break
                first.append(at); bases.append(count); geno.append(g)
        self._finish(np.array(first, np.int64), np.array(bases, np.int64),
                     np.array(geno, np.int8).reshape(-1, self.nsam))
        if len(self.first) == 0:
            raise NoDataError(path, self.data_start, int(window_end))

    def _finish(self, first, bases, genotype):
        p_first, p_bases, kind, src = split_long_rows(first, bases, self.max_segment_length)
        keep = p_first + p_bases > self.data_start         # pieces that end before the window are dropped
        self.first, self.bases, self.kind = p_first[keep], p_bases[keep], kind[keep]
        self.genotype = genotype[src[keep]]

    # ---- views
    def __len__(self):
        return len(self.first)

    @property
    def rows(self):
        """(first, bases, kind, genotype list) per row, file coordinates."""
        return [(int(a), int(b), int(k), [int(v) for v in g])
                for a, b, k, g in zip(self.first, self.bases, self.kind, self.genotype)]

    def pack(self, lags):
        """Arrays handed to the filter: coordinates relative to data_start (first base = 0)."""
        lo = np.maximum(0, self.first - self.data_start)
        hi = self.first + self.bases - self.data_start
        dist = distance_to_mutation(self.first, self.bases, self.genotype)
        if self.empty_file:
            dist[:] = 0.0                                  # pseudo rows never limit the recording
        mre = max_epoch_to_update_rows(lags, dist)
        return {"start": lo.astype(np.float64), "length": (hi - lo).astype(np.float64), "state": self.kind.copy(),
                "alleles": self.genotype.copy(), "max_record_epoch": mre, "distance_to_mutation": dist}

    def lookahead(self):
        return pack_lookahead(self, self.nsam)


def distance_to_mutation(fstart, flen, alleles):
    """Per row: 0 if it carries data, else the smaller of the distance back to the first row of its data-free run and
    forward to the end of the next row with data."""
    fstart = np.asarray(fstart, np.int64); flen = np.asarray(flen, np.int64)
    n = len(fstart)
    blank = (np.asarray(alleles) == -1).all(axis=1)
    idx = np.arange(n)
    # index of the first row of the current data-free run: last index that is blank with a non-blank predecessor
    run_open = blank & ~np.concatenate(([False], blank[:-1]))
    run_first = np.maximum.accumulate(np.where(run_open, idx, 0))
    back = (fstart - fstart[run_first]).astype(np.float64)
    # end of the next row with data at or after each row
    ends = np.where(~blank, fstart + flen, np.iinfo(np.int64).max)
    nxt_end = np.minimum.accumulate(ends[::-1])[::-1] if n else ends
    # minimum.accumulate picks the smallest end, which is the nearest because ends increase along the file
    fwd = np.where(nxt_end == np.iinfo(np.int64).max, np.inf, (nxt_end - fstart).astype(np.float64))
    return np.where(blank, np.minimum(back, fwd), 0.0)


def max_epoch_to_update(lags, distance):
    """Last epoch of the leading run whose half lag still exceeds the distance to data (-1: none)."""
    last = -1
    for e, lag in enumerate(lags):
        if not distance < 0.5 * lag:
            break
        last = e
    return last


def max_epoch_to_update_rows(lags, dist):
    ok = np.asarray(dist)[:, None] < 0.5 * np.asarray(lags, float)[None, :]
    lead = np.cumprod(ok, axis=1).sum(axis=1)
    return (lead - 1).astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------
# Look-ahead of the auxiliary particle filter (what Segment::set_lookahead, segdata.cpp:225-410, yields per row)

GIVE_UP_MISSING = 2000000.0          # a sample without data for this many bases is not waited for
TBL_QUANTILES = (0.001, 0.003, 0.01, 0.03, 0.1, 0.5, 0.95)   # smcsmc.cpp:134


class _SiteClass:
    """What the scan needs to know about one genotype row."""
    __slots__ = ("carriers", "absent", "first", "second", "unphased", "no_data")

    def __init__(self, g):
        n = len(g)
        self.carriers = 0; self.absent = 0; self.first = -1; self.second = -1
        self.unphased = [False] * n
        self.no_data = []
        j = 0
        while j < n:
            if g[j] > 0:
                self.carriers += 1
                if self.carriers == 1:
                    self.first = j
                elif self.carriers == 2:
                    self.second = j
                if g[j] == 2:                 # unphased pair: the partner haplotype is not looked at as a carrier
                    self.unphased[j] = True
                    j += 1
            if j < n and g[j] == -1:          # sees the partner of an unphased pair
                self.absent += 1
                self.no_data.append(j)
            j += 1


def _as_table(rows, nsam):
    if isinstance(rows, Segments):
        return rows.first, rows.bases, rows.genotype
    first = np.array([r[0] for r in rows], np.int64)
    bases = np.array([r[1] for r in rows], np.int64)
    geno = np.array([r[3] for r in rows], np.int8).reshape(len(rows), nsam)
    return first, bases, geno


def _scan(first, bases, geno, cls, here, nsam):
    next_private = [0.0] * nsam
    data_share = [0.0] * nsam
    pairs = []                       # [a, b, first_seen, last_seen, a_unphased, b_unphased, refuted]
    in_pair = [False] * (nsam + 1)
    split_at, split_row, split_minor = -1, None, 0
    have_private = have_private_unphased = samples_in_pairs = 0
    seen = with_data = 0.1
    streak = last_private = reach = 0.0
    origin = int(first[here])
    last_cls = cls[here]
    for i in range(here, len(first)):
        c = cls[i]
        last_cls = c
        nb = int(bases[i])
        if c.absent:
            streak += nb
            if streak > GIVE_UP_MISSING:
                for j in c.no_data:
                    if next_private[j] == 0:
                        mark = -(int(first[i]) - origin) - 1e-6
                        last_private = -mark
                        if mark < 0.5 * streak:
                            mark = -1e-6
                        next_private[j] = mark
                        data_share[j] = with_data / seen
                        have_private += 1
                    if not in_pair[j]:
                        in_pair[j] = True
                        samples_in_pairs += 1
        else:
            streak = 0.0
        seen += nb * nsam
        with_data += nb * (nsam - c.absent)
        if streak > GIVE_UP_MISSING:
            continue
        reach = int(first[i]) + nb - origin + 0.5
        g = geno[i]
        if c.carriers == 1:
            j = c.first
            if next_private[j] == 0:
                next_private[j] = reach
                data_share[j] = with_data / seen
                last_private = reach
                have_private += 1
                if c.unphased[j]:
                    next_private[j + 1] = reach
                    data_share[j + 1] = data_share[j]
                    have_private += 1
                    have_private_unphased += 1
        else:
            known = False
            for p in pairs:
                ga, gb = int(g[p[0]]), int(g[p[1]])
                if ((p[0] | 1) == p[1] and ga == 2) or (ga + gb == 1 and (ga | gb) == 1):
                    p[6] = True
                if c.carriers == 2 and p[0] == c.first and p[1] == c.second:
                    known = True
                    if not p[6]:
                        p[3] = reach
            if c.carriers == 2 and not known and g[c.first] > -1 and g[c.second] > -1:
                ua, ub = int(g[c.first] == 2), int(g[c.second] == 2)
                free = [(da, db) for da in range(ua + 1) for db in range(ub + 1)
                        if not in_pair[c.first + da] and not in_pair[c.second + db]]
                if free:
                    da, db = free[0]
                    pairs.append([c.first, c.second, reach, reach, bool(ua), bool(ub), False])
                    in_pair[c.first + da] = True
                    in_pair[c.second + db] = True
                    samples_in_pairs += 2
        if split_at == -1 and c.carriers > 2 and nsam - c.carriers > 2:
            split_at, split_row, split_minor = reach, i, min(c.carriers, nsam - c.carriers)
        if have_private == nsam:
            if samples_in_pairs >= nsam - 1:
                break
            if reach > (2 + have_private_unphased) * last_private:
                break
    if have_private < nsam:
        for j in range(nsam):
            if next_private[j] == 0:
                next_private[j] = -reach
                data_share[j] = with_data / seen
    return dict(first_singleton_distance=next_private, relative_mutation_rate=data_share,
                is_singleton_unphased=list(last_cls.unphased), doubleton=pairs, first_split_distance=split_at,
                split_alleles=[int(v) for v in geno[split_row]] if split_row is not None else [0] * nsam,
                split_count=split_minor)


def set_lookahead(rows, cur, nsam):
    """Look-ahead summary at row `cur`: distance to the next mutation private to each sample (negative: none within
    that distance) with the share of bases that had data, the sample pairs that next share a mutation (first / last
    evidence), and the first split into two groups of at least three."""
    first, bases, geno = _as_table(rows, nsam)
    cls = [None] * len(first)
    for i in range(cur, len(first)):
        cls[i] = _SiteClass(geno[i])
    return _scan(first, bases, geno, cls, cur, nsam)


def pack_lookahead(rows, nsam):
    """Per-row look-ahead arrays in the layout of pf_lookahead (include/smcsmc_pf.h)."""
    first, bases, geno = _as_table(rows, nsam)
    S = len(first)
    D = max(1, nsam // 2)
    out = dict(first_singleton_distance=np.zeros((S, nsam)), relative_mutation_rate=np.zeros((S, nsam)),
               is_singleton_unphased=np.zeros((S, nsam), np.int8), n_doubletons=np.zeros(S, np.int32),
               doubleton_idx=np.zeros((S, D, 4), np.int8), doubleton_dist=np.zeros((S, D, 2)),
               first_split_distance=np.full(S, -1.0), split_alleles=np.zeros((S, nsam), np.int8),
               split_count=np.zeros(S, np.int32), max_doubletons=D)
    cls = [_SiteClass(g) for g in geno]
    for i in range(S):
        la = _scan(first, bases, geno, cls, i, nsam)
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
