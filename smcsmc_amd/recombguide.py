"""From the local recombination map of one EM iteration (`<prefix>.recomb.gz`) to the recombination guide of the next
(`-guide`): the front-end's smoother, smcsmc/processrecombination.py:17-230 (`LocalRecombination`, called from
`Smcsmc.do_iteration`, model.py:1129-1143, when `-alpha > 0`).

Same interface (`LocalRecombination(infile)`, `.smooth(alpha, beta)`, `.write_data(outfile)`) and the same algorithm:
the per-window posterior rate minus the overall rate, change points by one pass of wild binary segmentation over a fixed
family of test windows (lengths 2 .. 2000 windows at half-length stride) accepted in order of their CUSUM contrast while
it exceeds beta x rate, a denser set of change points for the per-sample rates, piecewise means between change points,
mixed with the flat prior: alpha * posterior + (1 - alpha) * rate / n.

Differences from the reference's file, which is Python-2 code that no longer runs on the `.recomb.gz` its own binary
writes (its header says so: "will not work properly with the newfangled output files that include time-weighted
counts"): the two trailing columns `time`, `log_time` are not samples; the first locus is the chunk's start position
(1-based by default) and is rebased to 0, which is what a guide file must start at (pfparam.hpp:171-172); windows
without opportunity count as windows at the overall rate; the contrast of every test window is evaluated with numpy
on a prefix sum instead of a Python loop (identical arithmetic per window: sums of the same terms in the same order
are replaced by differences of prefix sums, so ties between near-equal contrasts may resolve differently)."""
import bisect
import gzip

import numpy as np

TEST_LENGTHS = (2, 3, 4, 6, 9, 13, 20, 30, 40, 60, 90, 130, 200, 300, 400, 600, 900, 1300, 2000)   # processrecombination.py:166-186


class LocalRecombination:
    def __init__(self, infile=None, opp=None, counts=None, step=100, iteration=0):
        """`infile`: a `.recomb` / `.recomb.gz` table; or `opp` [W] per-nt opportunity and `counts` [n][W] per-nt
        counts per sample (what ParticleFilter.local_recomb() returns, cumulated and divided by the interval)."""
        if infile is not None:
            opp, counts, step = self._read_data(infile, iteration)
        self.opp = np.asarray(opp, dtype=np.float64)
        self.counts = np.asarray(counts, dtype=np.float64)
        self.step = int(step)
        self.leaves = self.counts.shape[0]
        self.size = len(self.opp) * self.step
        total_opp = self.opp.sum()
        if not total_opp > 0:
            raise ValueError("Local recombination map holds no opportunity")
        self.rate = self.counts.sum() / total_opp                       # _calculate_rate, processrecombination.py:24-29
        self.smoothed_data = None

    @classmethod
    def from_filter(cls, local, nsam, interval=100.0):
        """From the arrays of pf_get_local_recomb: the differential opportunity is cumulated as
        CountModel::dump_local_recomb_logs does (count.cpp:632-636)."""
        opp = np.cumsum(np.asarray(local["opp_diff"], dtype=np.float64)) / interval
        counts = np.asarray(local["counts"], dtype=np.float64)[:nsam] / interval
        return cls(opp=opp, counts=counts, step=int(interval))

    @staticmethod
    def _read_data(infile, iteration=0):
        opener = gzip.open if infile.upper().endswith(".GZ") else open
        opp, counts, sizes = [], [], []
        nsam = None
        curpos = None
        with opener(infile, "rt") as f:
            for line in f:
                if line.startswith("iter"):
                    header = line.split()
                    nsam = len(header) - 4 - (2 if header[-1] == "log_time" else 0)
                    continue
                elts = line.split()
                it, locus, size = int(elts[0]), int(elts[1]), int(elts[2])
                if it < iteration:
                    continue
                if it > iteration:
                    break
                if curpos is not None and locus != curpos:
                    raise ValueError("Found gaps or overlaps in input file, line '{}'".format(line.strip()))
                curpos = locus + size
                vals = [float(v) for v in elts[3:]]
                if nsam is None:
                    nsam = len(vals) - 1
                opp.append(vals[0]); counts.append(vals[1:1 + nsam]); sizes.append(size)
        if not opp:
            raise ValueError("No records for iteration {} in {}".format(iteration, infile))
        step = int(np.gcd.reduce(np.array(sizes)))
        rep = np.array(sizes) // step                                   # _unmerge_data_generator: uniform windows
        return np.repeat(np.array(opp), rep), np.repeat(np.array(counts), rep, axis=0).T, step

    # ------------------------------------------------------------------------------------------------------------
    def _ratio(self, leaf=None):
        num = self.counts.sum(0) if leaf is None else self.counts[leaf]
        flat = self.rate if leaf is None else self.rate / self.leaves
        safe = np.where(self.opp > 0, self.opp, 1.0)
        return np.where(self.opp > 0, num / safe, flat)

    def _cusum(self, leaf=None):
        """processrecombination.py:31-42"""
        flat = self.rate if leaf is None else self.rate / self.leaves
        return np.cumsum(self._ratio(leaf) - flat)

    @staticmethod
    def _contrasts(C, s, l):
        """max_b |X^b_{s,s+l}| and its argmax for the windows [s, s+l) (argmax_xbse, processrecombination.py:135-157);
        C = prefix sums with C[0] = 0; s = array of window starts."""
        b = np.arange(1, l)                                             # change point offset: segments [s,s+b), [s+b,s+l)
        n = float(l)
        f1 = np.sqrt((l - b) / (n * b))
        f2 = np.sqrt(b / (n * (l - b)))
        left = C[s[:, None] + b[None, :]] - C[s][:, None]
        right = (C[s + l] - C[s])[:, None] - left
        x = np.abs(f1[None, :] * left - f2[None, :] * right)
        k = np.argmax(x, axis=1)                                        # first maximum, as the strict > of the loop
        return x[np.arange(len(s)), k], s + 1 + k

    def _wbs(self, cusum, beta, B=None):
        """One pass of wild binary segmentation (processrecombination.py:160-214)."""
        B = [] if B is None else B
        N = len(cusum)
        C = np.concatenate(([0.0], cusum))
        vals, bks, ss, es = [], [], [], []
        for l in TEST_LENGTHS:
            s = np.arange(0, N, l // 2)
            s = s[s + l < N]
            if len(s) == 0:
                continue
            # bounded blocks: a row of l-1 contrasts per window
            blk = max(1, (1 << 22) // l)
            for i in range(0, len(s), blk):
                v, b = self._contrasts(C, s[i:i + blk], l)
                vals.append(v); bks.append(b); ss.append(s[i:i + blk]); es.append(s[i:i + blk] + l)
        for s, e in zip([0] + B, B + [N]):                              # the segments between known change points
            if e - s >= 2:
                v, b = self._contrasts(C, np.array([s]), e - s)
                vals.append(v); bks.append(b); ss.append(np.array([s])); es.append(np.array([e]))
        if not vals:
            return B
        vals = np.concatenate(vals); bks = np.concatenate(bks); ss = np.concatenate(ss); es = np.concatenate(es)
        keep = vals >= beta * self.rate
        vals, bks, ss, es = vals[keep], bks[keep], ss[keep], es[keep]
        order = np.lexsort((es, ss, bks, -vals))                        # heap order of (-value, b, s, e)
        for i in order:
            s, e, bk = int(ss[i]), int(es[i]), int(bks[i])
            if bisect.bisect_right(B, s) != bisect.bisect_left(B, e):   # holds a change point already
                continue
            bisect.insort(B, bk)
        return B

    def _smooth_column(self, B, leaf=None):
        """piecewise means of the per-window rate between change points (processrecombination.py:52-69)"""
        r = self._ratio(leaf)
        edges = np.array([0] + list(B) + [len(r)])
        sums = np.add.reduceat(r, edges[:-1])
        lens = np.diff(edges)
        return np.repeat(sums / lens, lens)

    def smooth(self, alpha, beta):
        """processrecombination.py:216-230"""
        assert 0 <= alpha <= 1
        assert beta > 0
        B = self._wbs(self._cusum(), beta)
        total = self._smooth_column(B)
        Bp = list(B)
        for leaf in range(self.leaves):
            Bp = self._wbs(self._cusum(leaf), beta, Bp)
        cols = np.stack([self._smooth_column(Bp, leaf) for leaf in range(self.leaves)])
        rel = cols / (cols.sum(0) + 1e-30)
        self.change_points = B
        self.leaf_change_points = Bp
        self.smoothed_data = alpha * (rel * total[None, :]) + (1 - alpha) * self.rate / self.leaves
        return self

    # ------------------------------------------------------------------------------------------------------------
    def segments(self):
        """Runs of equal smoothed values: (start window, end window, rate, relative leaf rates)."""
        d = self.smoothed_data
        change = np.flatnonzero((d[:, 1:] != d[:, :-1]).any(0)) + 1
        starts = np.concatenate(([0], change))
        ends = np.concatenate((change, [d.shape[1]]))
        out = []
        for a, b in zip(starts, ends):
            v = d[:, a]
            rate = v.sum()
            out.append((int(a), int(b), rate, v / (rate + 1e-30)))
        return out

    def write_data(self, outfile):
        """The guide file (processrecombination.py:105-133): `locus size recomb_rate 1 .. n`, tab separated, 0-based."""
        outfile.write("locus\tsize\trecomb_rate" + "".join("\t{}".format(k + 1) for k in range(self.leaves)) + "\n")
        for a, b, rate, rel in self.segments():
            outfile.write("{}\t{}\t{:9.3e}".format(a * self.step, (b - a) * self.step, rate)
                          + "".join("\t{:5.3f}".format(v) for v in rel) + "\n")

    def guide(self, rounded=True):
        """The same segments as the dict the device model takes (`guide=`); with `rounded` the numbers take the detour
        through the file's formats, so an in-process run sees exactly what a run from files sees."""
        pos, rates, leaf = [], [], []
        for a, b, rate, rel in self.segments():
            pos.append(float(a * self.step))
            if rounded:
                rates.append(float("{:9.3e}".format(rate)))
                leaf.append([float("{:5.3f}".format(v)) for v in rel])
            else:
                rates.append(float(rate)); leaf.append(list(rel))
        return dict(positions=np.array(pos), rates=np.array(rates), leaf_rates=np.array(leaf).reshape(len(pos), self.leaves))
