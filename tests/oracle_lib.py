"""ctypes binding to the CPU oracle (oracle/libsmc_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.  The product package smcsmc_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class Model(C.Structure):
    _fields_ = [
        ("n_epochs", C.c_int32), ("n_pops", C.c_int32), ("nsam", C.c_int32), ("flags", C.c_int32),
        ("loci_length", C.c_double), ("mutation_rate", C.c_double), ("recombination_rate", C.c_double),
        ("change_times", C.POINTER(C.c_double)), ("pop_sizes", C.POINTER(C.c_double)),
        ("mig_rates", C.POINTER(C.c_double)), ("single_mig", C.POINTER(C.c_double)),
        ("sample_pops", C.POINTER(C.c_int32)), ("record_flags", C.POINTER(C.c_int32)),
        ("lags", C.POINTER(C.c_double)),
        ("n_bias_heights", C.c_int32), ("delay_type", C.c_int32),
        ("bias_heights", C.POINTER(C.c_double)), ("bias_strengths", C.POINTER(C.c_double)),
        ("application_delays", C.POINTER(C.c_double)),
        ("vb_coal_counts", C.POINTER(C.c_double)), ("vb_mig_counts", C.POINTER(C.c_double)),
        ("n_rate_segments", C.c_int32), ("reserved2", C.c_int32),
        ("rate_positions", C.POINTER(C.c_double)), ("rate_values", C.POINTER(C.c_double)),
        ("leaf_rel_rates", C.POINTER(C.c_double)),
    ]


class Params(C.Structure):
    _fields_ = [("np", C.c_int64), ("ess_fraction", C.c_double), ("seed", C.c_uint64),
                ("max_trace_events", C.c_int32), ("mig_cap", C.c_int32), ("delay_cap", C.c_int32), ("delay_evict", C.c_int32)]


class Segments(C.Structure):
    _fields_ = [("n", C.c_int64), ("start", C.POINTER(C.c_double)), ("length", C.POINTER(C.c_double)),
                ("state", C.POINTER(C.c_int8)), ("alleles", C.POINTER(C.c_int8)),
                ("max_record_epoch", C.POINTER(C.c_int32))]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libsmc_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.smco_create.restype = C.c_void_p
        L.smco_create.argtypes = [C.POINTER(Model), C.POINTER(Params)]
        L.smco_destroy.argtypes = [C.c_void_p]
        L.smco_last_error.restype = C.c_char_p
        L.smco_init_prior.argtypes = [C.c_void_p, C.c_double]
        L.smco_load_lookahead.argtypes = [C.c_void_p, C.c_void_p]
        L.smco_terminal_branch_quantiles.argtypes = [C.POINTER(Model), C.c_uint64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.smco_run.argtypes = [C.c_void_p, C.POINTER(Segments)]
        L.smco_update_segment.argtypes = [C.c_void_p, C.POINTER(Segments), C.c_int64]
        L.smco_count.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.smco_resample.argtypes = [C.c_void_p, C.c_double]
        L.smco_finish.argtypes = [C.c_void_p]
        L.smco_num_segments_done.restype = C.c_int64
        L.smco_num_segments_done.argtypes = [C.c_void_p]
        L.smco_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.smco_get_resample_events.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.smco_get_particles.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.smco_get_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.smco_get_migrations.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int32]
        L.smco_enable_local_recomb.argtypes = [C.c_void_p]
        L.smco_enable_tree_recording.argtypes = [C.c_void_p]
        L.smco_sample_tree_events.restype = C.c_int64
        L.smco_sample_tree_events.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.smco_sample_tree_events_pops.restype = C.c_int64
        L.smco_sample_tree_events_pops.argtypes = [C.c_void_p] * 7 + [C.c_int64, C.POINTER(C.c_int64)]
        L.smco_get_local_recomb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.smco_logl.restype = C.c_double
        L.smco_logl.argtypes = [C.c_void_p]
        L.smco_get_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.smco_get_delay_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("smco_exp", "smco_log", "smco_fastexp"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]
        L.smco_median_survival.argtypes = [C.POINTER(Model), C.c_uint64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
        L.smco_uniform.restype = C.c_double
        L.smco_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]
        L.smco_canon_sum.restype = C.c_double
        L.smco_canon_sum.argtypes = [C.c_void_p, C.c_int64]
        L.smco_canon_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.smco_systematic.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def attach_bias(owner, cmodel, m):
    """Optional focused-sampling fields of the model struct (kept alive on `owner`)."""
    bh = m.get("bias_heights")
    if bh is None or len(bh) == 0:
        cmodel.n_bias_heights = 0
        return
    owner._bh = np.ascontiguousarray(bh, dtype=np.float64)
    owner._bs = np.ascontiguousarray(m["bias_strengths"], dtype=np.float64)
    owner._ad = np.ascontiguousarray(m["application_delays"], dtype=np.float64)
    assert len(owner._bs) == len(owner._bh) + 1 and len(owner._ad) == cmodel.n_epochs
    cmodel.n_bias_heights = len(owner._bh)
    cmodel.delay_type = int(m.get("delay_type", 0))
    cmodel.bias_heights = _dp(owner._bh)
    cmodel.bias_strengths = _dp(owner._bs)
    cmodel.application_delays = _dp(owner._ad)


def attach_structure(owner, cmodel, m, E, P):
    """Per-epoch migration matrix [E][P][P], fixed-time moves [E][P][P] and the samples' populations."""
    if m.get("mig_rates") is not None:
        owner._mig = np.ascontiguousarray(m["mig_rates"], dtype=np.float64).reshape(E * P * P)
        cmodel.mig_rates = _dp(owner._mig)
    if m.get("single_mig") is not None:
        owner._smig = np.ascontiguousarray(m["single_mig"], dtype=np.float64).reshape(E * P * P)
        cmodel.single_mig = _dp(owner._smig)
    if m.get("guide") is not None:
        # recombination guide: dict(positions[K], rates[K], leaf_rates[K][nsam]) (RecombinationBias, pfparam.hpp:96-223)
        gd = m["guide"]
        owner._gpos = np.ascontiguousarray(gd["positions"], dtype=np.float64)
        owner._grate = np.ascontiguousarray(gd["rates"], dtype=np.float64)
        owner._gleaf = np.ascontiguousarray(gd["leaf_rates"], dtype=np.float64).reshape(len(owner._gpos) * cmodel.nsam)
        cmodel.n_rate_segments = len(owner._gpos)
        cmodel.rate_positions = _dp(owner._gpos)
        cmodel.rate_values = _dp(owner._grate)
        cmodel.leaf_rel_rates = _dp(owner._gleaf)
        if not cmodel.application_delays:
            owner._gad = np.ascontiguousarray(m["application_delays"], dtype=np.float64)
            cmodel.application_delays = _dp(owner._gad)
            cmodel.delay_type = int(m.get("delay_type", 0))
    if m.get("vb_coal_counts") is not None:
        # variational-Bayes event counts: [E][P] per coalescence, [E][P][P] per migration (-vb)
        owner._vbc = np.ascontiguousarray(m["vb_coal_counts"], dtype=np.float64).reshape(E * P)
        cmodel.vb_coal_counts = _dp(owner._vbc)
        if m.get("vb_mig_counts") is not None:
            owner._vbm = np.ascontiguousarray(m["vb_mig_counts"], dtype=np.float64).reshape(E * P * P)
            cmodel.vb_mig_counts = _dp(owner._vbm)
    if m.get("sample_pops") is not None:
        owner._spop = np.ascontiguousarray(m["sample_pops"], dtype=np.int32)
        assert len(owner._spop) == cmodel.nsam
        cmodel.sample_pops = owner._spop.ctypes.data_as(C.POINTER(C.c_int32))


class PackedInputs:
    """Owns the numpy buffers behind the C structs (same layout for oracle and product)."""

    def __init__(self, model, segs, model_cls=Model, seg_cls=Segments):
        m = model
        self.change_times = np.ascontiguousarray(m["change_times"], dtype=np.float64)
        E = len(self.change_times)
        P = int(m.get("n_pops", 1))
        self.pop_sizes = np.ascontiguousarray(m["pop_sizes"], dtype=np.float64).reshape(E * P)
        self.record_flags = np.ascontiguousarray(m.get("record_flags", [3] * E), dtype=np.int32)
        self.lags = np.ascontiguousarray(m["lags"], dtype=np.float64)
        flags = (1 if m.get("ancestral_aware") else 0) | (2 if m.get("dephase") else 0)
        self.E, self.P, self.nsam = E, P, int(m["nsam"])
        self.model = model_cls(E, P, int(m["nsam"]), flags, float(m["loci_length"]), float(m["mutation_rate"]),
                               float(m["recombination_rate"]), _dp(self.change_times), _dp(self.pop_sizes),
                               None, None, None,
                               self.record_flags.ctypes.data_as(C.POINTER(C.c_int32)), _dp(self.lags))
        attach_structure(self, self.model, m, E, P)
        attach_bias(self, self.model, m)
        self.segs = None
        if segs is not None:
            self.start = np.ascontiguousarray(segs["start"], dtype=np.float64)
            self.length = np.ascontiguousarray(segs["length"], dtype=np.float64)
            self.state = np.ascontiguousarray(segs["state"], dtype=np.int8)
            self.alleles = np.ascontiguousarray(segs["alleles"], dtype=np.int8).reshape(-1)
            self.mre = np.ascontiguousarray(segs["max_record_epoch"], dtype=np.int32)
            n = len(self.start)
            assert len(self.alleles) == n * self.nsam
            self.segs = seg_cls(n, _dp(self.start), _dp(self.length),
                                self.state.ctypes.data_as(C.POINTER(C.c_int8)),
                                self.alleles.ctypes.data_as(C.POINTER(C.c_int8)),
                                self.mre.ctypes.data_as(C.POINTER(C.c_int32)))


class Oracle:
    def __init__(self, model, np_particles, ess_fraction=0.5, seed=1, max_trace_events=64, mig_cap=0, delay_cap=0,
                 delay_evict=False):
        self.L = lib()
        self.inp = PackedInputs(model, None)
        self.Np = int(np_particles)
        self.params = Params(self.Np, float(ess_fraction), int(seed), int(max_trace_events), int(mig_cap), int(delay_cap),
                             int(bool(delay_evict)))
        self.mig_cap = int(mig_cap) if mig_cap else 96
        self.h = self.L.smco_create(C.byref(self.inp.model), C.byref(self.params))
        if not self.h:
            raise RuntimeError(self.L.smco_last_error().decode())
        self.max_trace_events = max_trace_events

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.smco_last_error().decode())
        return rc

    def close(self):
        if self.h:
            self.L.smco_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def init_prior(self, initial_position=0.0):
        self._chk(self.L.smco_init_prior(self.h, float(initial_position)))

    def load_lookahead(self, la, level, tbl, quantiles=None):
        from smcsmc_amd import pf, segments as segmod      # struct layout only (shared with the product's C-ABI)
        self._la = pf.PackedLookahead(la, level, tbl, segmod.TBL_QUANTILES if quantiles is None else quantiles)
        self._chk(self.L.smco_load_lookahead(self.h, C.byref(self._la.struct)))

    def pack_segments(self, model, segs):
        self.seg_inp = PackedInputs(model, segs)
        return self.seg_inp

    def run(self, seg_inp):
        self._chk(self.L.smco_run(self.h, C.byref(seg_inp.segs)))

    def enable_tree_recording(self):
        """-arg: before init_prior"""
        self._chk(self.L.smco_enable_tree_recording(self.h))

    def sample_tree_events(self, pops=False):
        part = C.c_int64()
        n = self.L.smco_sample_tree_events_pops(self.h, None, None, None, None, None, None, 0, C.byref(part))
        if n < 0:
            raise RuntimeError("tree recording is off")
        kind = np.zeros(n, np.int32); pos = np.zeros(n); hgt = np.zeros(n); desc = np.zeros(n, np.uint32)
        fr = np.zeros(n, np.int32); to = np.zeros(n, np.int32)
        self.L.smco_sample_tree_events_pops(self.h, kind.ctypes.data, pos.ctypes.data, hgt.ctypes.data, desc.ctypes.data,
                                            fr.ctypes.data, to.ctypes.data, n, C.byref(part))
        if pops:
            return int(part.value), kind, pos, hgt, desc, fr, to
        return int(part.value), kind, pos, hgt, desc

    def update_segment(self, seg_inp, s):
        self._chk(self.L.smco_update_segment(self.h, C.byref(seg_inp.segs), s))

    def count(self, pos, end_data=False):
        self._chk(self.L.smco_count(self.h, float(pos), int(end_data)))

    def resample(self, pos):
        return self._chk(self.L.smco_resample(self.h, float(pos)))

    def finish(self):
        self._chk(self.L.smco_finish(self.h))

    def trace(self):
        n = self.L.smco_num_segments_done(self.h)
        T = np.zeros(n); ess = np.zeros(n); flag = np.zeros(n, np.int32); logl = np.zeros(n)
        self.L.smco_get_trace(self.h, T.ctypes.data, ess.ctypes.data, flag.ctypes.data, logl.ctypes.data, n)
        return {"T": T, "ess": ess, "resampled": flag, "logl": logl}

    def resample_events(self):
        seg = np.zeros(self.max_trace_events, np.int32)
        par = np.zeros((self.max_trace_events, self.Np), np.int32)
        n = self.L.smco_get_resample_events(self.h, seg.ctypes.data, par.ctypes.data, self.max_trace_events)
        return seg[:n], par[:n]

    def particles(self):
        n = self.inp.nsam
        wp = np.zeros(self.Np); wq = np.zeros(self.Np); H = np.zeros((self.Np, n - 1))
        Ch = np.zeros((self.Np, n - 1, 2), np.int8); nb = np.zeros(self.Np)
        self.L.smco_get_particles(self.h, wp.ctypes.data, wq.ctypes.data, H.ctypes.data, Ch.ctypes.data, nb.ctypes.data)
        return {"w_post": wp, "w_pilot": wq, "heights": H, "children": Ch, "next_base": nb}

    def counts(self):
        E, P = self.inp.E, self.inp.P
        out = np.zeros(counts_len(E, P))
        self._chk(self.L.smco_get_counts(self.h, out.ctypes.data, len(out)))
        return unpack_counts(out, E, P)

    def enable_local_recomb(self):
        self.L.smco_enable_local_recomb(self.h)

    def local_recomb(self, loci_length):
        nb = int(loci_length / 100.0)
        opp = np.zeros(nb); cnt = np.zeros((self.inp.nsam + 2, nb))
        self._chk(self.L.smco_get_local_recomb(self.h, opp.ctypes.data, cnt.ctypes.data, nb))
        return {"opp_diff": opp, "counts": cnt}

    def migrations(self, cap=None):
        n = self.inp.nsam
        cap = self.mig_cap if cap is None else cap
        nm = np.zeros(self.Np, np.int32); t = np.zeros((self.Np, cap)); b = np.zeros((self.Np, cap), np.int8)
        q = np.zeros((self.Np, cap), np.int8); npop = np.zeros((self.Np, n - 1), np.int8)
        self.L.smco_get_migrations(self.h, nm.ctypes.data, t.ctypes.data, b.ctypes.data, q.ctypes.data,
                                   npop.ctypes.data, cap)
        return {"n_events": nm, "times": t, "branch": b, "newpop": q, "node_pops": npop}

    def logl(self):
        return self.L.smco_logl(self.h)

    def stats(self):
        a = C.c_int64(); b = C.c_int64(); c = C.c_int64()
        self.L.smco_get_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"recombinations": a.value, "events_allocated": b.value, "resamples": c.value}

    def delay_stats(self):
        """(factors applied ahead of their position to make room, most factors any particle had pending)"""
        a = C.c_int64(); b = C.c_int32()
        self.L.smco_get_delay_stats(self.h, C.byref(a), C.byref(b))
        return {"forced": a.value, "peak": b.value}


def counts_len(E, P=1):
    return 6 * E + 4 if P == 1 else 3 * E * P + 3 * E + E * P * P + 2 * E * P + 4


def unpack_counts(out, E, P=1):
    """Packed count buffer -> dict (layout: include/smcsmc_pf.h PF_COUNTS_LEN / PF_COUNTS_LEN2)."""
    if P == 1:
        return {
            "coal_count": out[0:E].copy(), "coal_opp": out[E:2 * E].copy(), "coal_weight": out[2 * E:3 * E].copy(),
            "rec_count": out[3 * E:4 * E].copy(), "rec_opp": out[4 * E:5 * E].copy(), "rec_weight": out[5 * E:6 * E].copy(),
            "delayed_opp": out[6 * E], "delayed_count": out[6 * E + 1], "resample_count": out[6 * E + 2],
            "logl": out[6 * E + 3],
        }
    o = 0
    d = {}
    for k in ("coal_count", "coal_opp", "coal_weight"):
        d[k] = out[o:o + E * P].reshape(E, P).copy(); o += E * P
    for k in ("rec_count", "rec_opp", "rec_weight"):
        d[k] = out[o:o + E].copy(); o += E
    d["mig_count"] = out[o:o + E * P * P].reshape(E, P, P).copy(); o += E * P * P
    for k in ("mig_opp", "mig_weight"):
        d[k] = out[o:o + E * P].reshape(E, P).copy(); o += E * P
    d["delayed_opp"], d["delayed_count"], d["resample_count"], d["logl"] = out[o:o + 4]
    return d


def terminal_branch_quantiles(model, seed=1, n_trees=100000, quantiles=None):
    from smcsmc_amd import segments as segmod
    L = lib()
    inp = PackedInputs(dict(model, lags=model.get("lags", np.zeros(len(model["change_times"])))), None)
    q = np.ascontiguousarray(segmod.TBL_QUANTILES if quantiles is None else quantiles, dtype=np.float64)
    out = np.zeros((inp.nsam, len(q)))
    mean = C.c_double()
    if L.smco_terminal_branch_quantiles(C.byref(inp.model), int(seed), int(n_trees), q.ctypes.data, len(q), out.ctypes.data,
                                        C.byref(mean)) < 0:
        raise RuntimeError(L.smco_last_error().decode())
    return out, mean.value


def median_survival(model, seed=1, min_events=200, max_trees=1000000):
    L = lib()
    inp = PackedInputs(dict(model, lags=model.get("lags", np.zeros(len(model["change_times"])))), None)
    out = np.zeros(inp.E)
    trees = C.c_int64()
    if L.smco_median_survival(C.byref(inp.model), int(seed), int(min_events), int(max_trees), out.ctypes.data,
                              C.byref(trees)) < 0:
        raise RuntimeError(L.smco_last_error().decode())
    return out, trees.value
