"""ctypes binding to the HIP particle-filter library (C-ABI: include/smcsmc_pf.h).

Mirrors the reference's ParticleContainer / CountModel call surface
(/root/reference/src/particleContainer.hpp:48-70, count.hpp:52-63):

    pf = ParticleFilter(model, Np, ess_fraction, seed)
    pf.init_prior(x0)            # ParticleContainer ctor
    pf.load_segments(segs)       # Segment buffer
    pf.run()                     # the pfARG_core do-while (smcsmc.cpp:324-360)
    pf.finish()                  # smcsmc.cpp:371-373
    pf.counts(), pf.logl()

Raises PfError (the reference prints "Error: ..." and exits 1) -- in particular when the HIP
library or a GPU is missing: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsmcsmc_pf.so")
_LIB = None


class PfError(RuntimeError):
    pass


class _Model(C.Structure):
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


class _Params(C.Structure):
    _fields_ = [("np", C.c_int64), ("ess_fraction", C.c_double), ("seed", C.c_uint64),
                ("max_trace_events", C.c_int32), ("flags", C.c_int32),
                ("log_cap", C.c_int64), ("gen_cap", C.c_int64), ("piece_cap", C.c_int64),
                ("debug", C.c_int32), ("mig_cap", C.c_int32), ("count_wgs", C.c_int32), ("delay_cap", C.c_int32),
                ("count_workers", C.c_int32), ("reserved4", C.c_int32)]


DEBUG_FORCE_LDS, DEBUG_NO_FUSE, DEBUG_NO_COUNT, DEBUG_TWO_LAUNCH = 1, 2, 4, 8


class _Segments(C.Structure):
    _fields_ = [("n", C.c_int64), ("start", C.POINTER(C.c_double)), ("length", C.POINTER(C.c_double)),
                ("state", C.POINTER(C.c_int8)), ("alleles", C.POINTER(C.c_int8)),
                ("max_record_epoch", C.POINTER(C.c_int32))]


class _Lookahead(C.Structure):
    _fields_ = [("level", C.c_int32), ("max_doubletons", C.c_int32), ("n_quantiles", C.c_int32), ("reserved", C.c_int32),
                ("n", C.c_int64),
                ("first_singleton_distance", C.POINTER(C.c_double)), ("relative_mutation_rate", C.POINTER(C.c_double)),
                ("is_singleton_unphased", C.POINTER(C.c_int8)), ("n_doubletons", C.POINTER(C.c_int32)),
                ("doubleton_idx", C.POINTER(C.c_int8)), ("doubleton_dist", C.POINTER(C.c_double)),
                ("first_split_distance", C.POINTER(C.c_double)), ("split_alleles", C.POINTER(C.c_int8)),
                ("split_count", C.POINTER(C.c_int32)), ("quantiles", C.POINTER(C.c_double)),
                ("tbl_lengths", C.POINTER(C.c_double)), ("mean_total_branch_length", C.c_double)]


class PackedLookahead:
    """Owns the buffers behind a pf_lookahead / smco_lookahead struct.  `la` = segments.pack_lookahead(...),
    `tbl` = (lengths[nsam][Q], mean_total_branch_length) from terminal_branch_quantiles."""

    def __init__(self, la, level, tbl, quantiles, struct_cls=_Lookahead):
        f = lambda a, t: np.ascontiguousarray(a, dtype=t)      # noqa: E731
        self.a = [f(la["first_singleton_distance"], np.float64), f(la["relative_mutation_rate"], np.float64),
                  f(la["is_singleton_unphased"], np.int8), f(la["n_doubletons"], np.int32), f(la["doubleton_idx"], np.int8),
                  f(la["doubleton_dist"], np.float64), f(la["first_split_distance"], np.float64),
                  f(la["split_alleles"], np.int8), f(la["split_count"], np.int32), f(quantiles, np.float64),
                  f(tbl[0], np.float64)]
        ptr = lambda a: a.ctypes.data_as(C.POINTER({np.dtype(np.float64): C.c_double, np.dtype(np.int8): C.c_int8,      # noqa: E731
                                                    np.dtype(np.int32): C.c_int32}[a.dtype]))
        self.struct = struct_cls(int(level), int(la["max_doubletons"]), len(quantiles), 0, len(la["n_doubletons"]),
                                 *[ptr(a) for a in self.a], float(tbl[1]))


EXPORTS = [
    "pf_last_error", "pf_device_count", "pf_create", "pf_destroy", "pf_init_prior", "pf_load_segments", "pf_load_lookahead",
    "pf_terminal_branch_quantiles",
    "pf_update_segment", "pf_count", "pf_resample", "pf_run", "pf_run_many", "pf_can_run_many", "pf_finish", "pf_sync",
    "pf_num_segments_done", "pf_logl", "pf_get_counts", "pf_get_trace", "pf_get_resample_events",
    "pf_get_particles", "pf_get_migrations", "pf_get_local_recomb", "pf_sample_tree_events", "pf_sample_tree_events_pops", "pf_get_kernel_time", "pf_set_timing", "pf_get_stats", "pf_get_delay_stats", "pf_probe_handoff", "pf_set_wg_trace", "pf_get_wg_trace", "pf_debug_stamps", "pf_test_search_lut", "pf_simulate_sites",
    "pf_median_survival", "pf_test_math", "pf_test_div", "pf_test_uniform", "pf_test_reduce", "pf_test_systematic",
]


def load_library(path=None):
    """Loads the HIP library; raises PfError loudly if it has not been built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise PfError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.pf_last_error.restype = C.c_char_p
    L.pf_device_count.restype = C.c_int
    L.pf_create.restype = vp
    L.pf_create.argtypes = [C.POINTER(_Model), C.POINTER(_Params), C.c_int]
    L.pf_destroy.argtypes = [vp]
    L.pf_init_prior.argtypes = [vp, C.c_double]
    L.pf_load_segments.argtypes = [vp, C.POINTER(_Segments)]
    L.pf_load_lookahead.argtypes = [vp, C.POINTER(_Lookahead)]
    L.pf_terminal_branch_quantiles.argtypes = [C.POINTER(_Model), C.c_uint64, C.c_int64, vp, C.c_int32, vp, vp, C.c_int]
    L.pf_update_segment.argtypes = [vp, C.c_int64]
    L.pf_count.argtypes = [vp, C.c_int64, C.c_int]
    L.pf_resample.argtypes = [vp, C.c_int64]
    L.pf_run.argtypes = [vp, C.c_int64, C.c_int64]
    L.pf_run_many.argtypes = [vp, C.c_int32, C.c_int64, C.c_int64]
    L.pf_can_run_many.argtypes = [vp, C.c_int32]
    L.pf_finish.argtypes = [vp]
    L.pf_sync.argtypes = [vp]
    L.pf_sample_tree_events.restype = C.c_int64
    L.pf_sample_tree_events.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.pf_sample_tree_events_pops.restype = C.c_int64
    L.pf_sample_tree_events_pops.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.pf_num_segments_done.restype = C.c_int64
    L.pf_num_segments_done.argtypes = [vp]
    L.pf_logl.restype = C.c_double
    L.pf_logl.argtypes = [vp]
    L.pf_get_counts.argtypes = [vp, vp, C.c_int32]
    L.pf_get_trace.argtypes = [vp, vp, vp, vp, vp, C.c_int64]
    L.pf_get_resample_events.argtypes = [vp, vp, vp, C.c_int32]
    L.pf_get_particles.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pf_get_migrations.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32]
    L.pf_get_local_recomb.argtypes = [vp, vp, vp, C.c_int64]
    L.pf_get_kernel_time.argtypes = [vp, C.c_int, vp, vp]
    L.pf_set_timing.argtypes = [vp, C.c_int]
    L.pf_get_stats.argtypes = [vp, vp, vp, vp]
    L.pf_get_delay_stats.argtypes = [vp, vp, vp]
    L.pf_probe_handoff.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, vp, vp, C.c_int32]
    L.pf_set_wg_trace.argtypes = [vp, C.c_int64, C.c_int32]
    L.pf_get_wg_trace.argtypes = [vp, vp, C.c_int64, vp]
    L.pf_get_wg_trace.restype = C.c_int64
    L.pf_median_survival.argtypes = [C.POINTER(_Model), C.c_uint64, C.c_int32, C.c_int64, vp, vp, C.c_int]
    L.pf_test_math.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int]
    L.pf_test_div.argtypes = [vp, vp, C.c_int64, vp, C.c_int]
    L.pf_test_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64, vp, C.c_int]
    L.pf_test_reduce.argtypes = [vp, C.c_int64, vp, vp, C.c_int]
    L.pf_test_systematic.argtypes = [vp, C.c_int64, C.c_double, vp, C.c_int]
    if path == LIB_PATH:
        _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _err(L):
    return (L.pf_last_error() or b"").decode()


def _attach_bias(owner, cmodel, m):
    """Focused sampling (-bias_heights / -bias_strengths) and the application delays of the delayed
    importance weights; absent keys switch the feature off."""
    bh = m.get("bias_heights")
    if bh is None or len(bh) == 0:
        cmodel.n_bias_heights = 0
        return
    owner._bh = np.ascontiguousarray(bh, dtype=np.float64)
    owner._bs = np.ascontiguousarray(m["bias_strengths"], dtype=np.float64)
    owner._ad = np.ascontiguousarray(m["application_delays"], dtype=np.float64)
    if len(owner._bs) != len(owner._bh) + 1 or len(owner._ad) != cmodel.n_epochs:
        raise PfError("bias_strengths needs one more entry than bias_heights; application_delays one per epoch")
    cmodel.n_bias_heights = len(owner._bh)
    cmodel.delay_type = int(m.get("delay_type", 0))
    cmodel.bias_heights = _dp(owner._bh)
    cmodel.bias_strengths = _dp(owner._bs)
    cmodel.application_delays = _dp(owner._ad)


def _attach_structure(owner, cmodel, m, E, P):
    """Structured models (scrm -I / -eM / -ema / -ej): migration matrix [E][P][P] (backward rate p -> q per
    generation), fixed-time moves [E][P][P] applied at the start of an epoch, and the samples' populations."""
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
        if len(owner._spop) != cmodel.nsam:
            raise PfError("sample_pops needs one entry per sample")
        cmodel.sample_pops = owner._spop.ctypes.data_as(C.POINTER(C.c_int32))


def counts_len(E, P=1):
    """PF_COUNTS_LEN2 of include/smcsmc_pf.h"""
    return 6 * E + 4 if P == 1 else 3 * E * P + 3 * E + E * P * P + 2 * E * P + 4


def unpack_counts(out, E, P=1):
    """Packed count buffer -> dict of the CountModel members (count.hpp:95-110)."""
    if P == 1:
        return {
            "coal_count": out[0:E].copy(), "coal_opp": out[E:2 * E].copy(), "coal_weight": out[2 * E:3 * E].copy(),
            "rec_count": out[3 * E:4 * E].copy(), "rec_opp": out[4 * E:5 * E].copy(),
            "rec_weight": out[5 * E:6 * E].copy(), "delayed_opp": out[6 * E], "delayed_count": out[6 * E + 1],
            "resample_count": out[6 * E + 2], "logl": out[6 * E + 3],
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
    d["delayed_opp"], d["delayed_count"], d["resample_count"], d["logl"] = (float(v) for v in out[o:o + 4])
    return d


KERNEL_CLASSES = ("extend", "decide", "count", "resample")


def probe_handoff(mode, rows=2000, nw=157, spin_us=0.0, device=0):
    """microseconds per row of the row hand-off probe (pf_probe_handoff): mode 0 = a kernel boundary per row, 1 = a resident grid"""
    L = load_library()
    us = C.c_double(); cs = C.c_double()
    rc = L.pf_probe_handoff(int(mode), int(rows), int(nw), int(round(spin_us * 100.0)), C.byref(us), C.byref(cs), int(device))
    if rc != 0:
        raise PfError("pf_probe_handoff failed (%d)" % rc)
    return us.value, cs.value


class ParticleFilter:
    def __init__(self, model, np_particles, ess_fraction=0.5, seed=1, max_trace_events=64, device=0, local_recomb=False,
                 record_trees=False, log_cap=0, gen_cap=0, piece_cap=0, debug=0, mig_cap=0, count_wgs=0, delay_cap=0,
                 delay_evict=False, count_workers=0):
        self.L = load_library()
        m = model
        self._ct = np.ascontiguousarray(m["change_times"], dtype=np.float64)
        E = len(self._ct)
        P = int(m.get("n_pops", 1))
        self._ps = np.ascontiguousarray(m["pop_sizes"], dtype=np.float64).reshape(E * P)
        self._rf = np.ascontiguousarray(m.get("record_flags", [3] * E), dtype=np.int32)
        self._lags = np.ascontiguousarray(m["lags"], dtype=np.float64)
        flags = (1 if m.get("ancestral_aware") else 0) | (2 if m.get("dephase") else 0)
        self.E, self.P, self.nsam, self.Np = E, P, int(m["nsam"]), int(np_particles)
        self.max_trace_events = int(max_trace_events)
        self._model = _Model(E, P, self.nsam, flags, float(m["loci_length"]), float(m["mutation_rate"]),
                             float(m["recombination_rate"]), _dp(self._ct), _dp(self._ps), None, None, None,
                             self._rf.ctypes.data_as(C.POINTER(C.c_int32)), _dp(self._lags))
        _attach_bias(self, self._model, m)
        _attach_structure(self, self._model, m, E, P)
        self.loci_length = float(m["loci_length"])
        self._params = _Params(self.Np, float(ess_fraction), int(seed), self.max_trace_events,
                               (1 if local_recomb else 0) | (2 if record_trees else 0) | (4 if delay_evict else 0),
                               int(log_cap), int(gen_cap), int(piece_cap), int(debug), int(mig_cap), int(count_wgs), int(delay_cap), int(count_workers), 0)
        self.mig_cap = int(mig_cap) if mig_cap else 96
        self.h = self.L.pf_create(C.byref(self._model), C.byref(self._params), int(device))
        if not self.h:
            raise PfError(_err(self.L))
        self.n_segs = 0

    def _chk(self, rc):
        if rc < 0:
            raise PfError(_err(self.L))
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.pf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_prior(self, initial_position=0.0):
        self._chk(self.L.pf_init_prior(self.h, float(initial_position)))

    def load_segments(self, segs):
        self._s0 = np.ascontiguousarray(segs["start"], dtype=np.float64)
        self._s1 = np.ascontiguousarray(segs["length"], dtype=np.float64)
        self._s2 = np.ascontiguousarray(segs["state"], dtype=np.int8)
        self._s3 = np.ascontiguousarray(segs["alleles"], dtype=np.int8).reshape(-1)
        self._s4 = np.ascontiguousarray(segs["max_record_epoch"], dtype=np.int32)
        n = len(self._s0)
        assert len(self._s3) == n * self.nsam
        sg = _Segments(n, _dp(self._s0), _dp(self._s1), self._s2.ctypes.data_as(C.POINTER(C.c_int8)),
                       self._s3.ctypes.data_as(C.POINTER(C.c_int8)), self._s4.ctypes.data_as(C.POINTER(C.c_int32)))
        self._chk(self.L.pf_load_segments(self.h, C.byref(sg)))
        self.n_segs = n

    def load_lookahead(self, la, level, tbl, quantiles=None):
        """Switches the auxiliary particle filter on (-apf level): `la` from segments.pack_lookahead, `tbl` from
        terminal_branch_quantiles.  Call after load_segments."""
        from . import segments as segmod
        self._la = PackedLookahead(la, level, tbl, segmod.TBL_QUANTILES if quantiles is None else quantiles)
        self._chk(self.L.pf_load_lookahead(self.h, C.byref(self._la.struct)))

    def run(self, s_begin=0, s_end=None):
        self._chk(self.L.pf_run(self.h, int(s_begin), int(self.n_segs if s_end is None else s_end)))

    @staticmethod
    def run_many(filters, s_begin=0, s_end=None):
        """Rows [s_begin, s_end) of several chunks (filters on one device, same shape) in lockstep: one launch per row
        covers all of them (pf_run_many).  A chunk with fewer rows stops at its end."""
        if s_end is None:
            s_end = max(f.n_segs for f in filters)
        hs = (C.c_void_p * len(filters))(*[f.h for f in filters])
        filters[0]._chk(filters[0].L.pf_run_many(hs, len(filters), int(s_begin), int(s_end)))

    def update_segment(self, s):
        self._chk(self.L.pf_update_segment(self.h, int(s)))

    def count(self, s, end_data=False):
        self._chk(self.L.pf_count(self.h, int(s), int(end_data)))

    def resample(self, s):
        self._chk(self.L.pf_resample(self.h, int(s)))

    def finish(self):
        self._chk(self.L.pf_finish(self.h))

    def sync(self):
        self._chk(self.L.pf_sync(self.h))

    def segments_done(self):
        return int(self.L.pf_num_segments_done(self.h))

    def logl(self):
        return float(self.L.pf_logl(self.h))

    def trace(self):
        self.sync()
        n = self.segments_done()
        T = np.zeros(n); ess = np.zeros(n); flag = np.zeros(n, np.int32); logl = np.zeros(n)
        self._chk(self.L.pf_get_trace(self.h, T.ctypes.data, ess.ctypes.data, flag.ctypes.data, logl.ctypes.data, n))
        return {"T": T, "ess": ess, "resampled": flag, "logl": logl}

    def resample_events(self):
        seg = np.zeros(max(1, self.max_trace_events), np.int32)
        par = np.zeros((max(1, self.max_trace_events), self.Np), np.int32)
        n = self._chk(self.L.pf_get_resample_events(self.h, seg.ctypes.data, par.ctypes.data, self.max_trace_events))
        return seg[:n], par[:n]

    def particles(self):
        n = self.nsam
        wp = np.zeros(self.Np); wq = np.zeros(self.Np); H = np.zeros((self.Np, n - 1))
        Ch = np.zeros((self.Np, n - 1, 2), np.int8); nb = np.zeros(self.Np)
        self._chk(self.L.pf_get_particles(self.h, wp.ctypes.data, wq.ctypes.data, H.ctypes.data, Ch.ctypes.data,
                                          nb.ctypes.data))
        return {"w_post": wp, "w_pilot": wq, "heights": H, "children": Ch, "next_base": nb}

    def counts(self):
        out = np.zeros(counts_len(self.E, self.P))
        self._chk(self.L.pf_get_counts(self.h, out.ctypes.data, len(out)))
        return unpack_counts(out, self.E, self.P)

    def local_recomb(self):
        """The 100-bp local recombination map: differential opportunity [nbins], counts [nsam+2][nbins]
        (per sample, time-weighted, log-time-weighted); nbins = loci_length / 100 as dump_local_recomb_logs writes."""
        nb = int(self.loci_length / 100.0)
        opp = np.zeros(nb); cnt = np.zeros((self.nsam + 2, nb))
        self._chk(self.L.pf_get_local_recomb(self.h, opp.ctypes.data, cnt.ctypes.data, nb))
        return {"opp_diff": opp, "counts": cnt}

    def sample_tree_events(self, pops=False):
        """-arg: the particle of the final one-particle draw and the tree-modifying events of its history, last position
        first: (particle, kind[K] (0 R, 1 C, 2 M), pos[K], height[K], descendants[K] as sample bit masks); with pops=True
        also (from_pop[K], to_pop[K]), the population columns of the .trees.gz lines."""
        part = C.c_int64()
        n = self.L.pf_sample_tree_events_pops(self.h, None, None, None, None, None, None, 0, C.byref(part))
        if n < 0:
            raise PfError(_err(self.L))
        kind = np.zeros(n, np.int32); pos = np.zeros(n); hgt = np.zeros(n); desc = np.zeros(n, np.uint32)
        fr = np.zeros(n, np.int32); to = np.zeros(n, np.int32)
        self.L.pf_sample_tree_events_pops(self.h, kind.ctypes.data, pos.ctypes.data, hgt.ctypes.data, desc.ctypes.data,
                                          fr.ctypes.data, to.ctypes.data, n, C.byref(part))
        if pops:
            return int(part.value), kind, pos, hgt, desc, fr, to
        return int(part.value), kind, pos, hgt, desc

    def migrations(self, cap=None):
        """Migration events on every particle's local tree and the population of every coalescent node."""
        n = self.nsam
        cap = self.mig_cap if cap is None else cap
        nm = np.zeros(self.Np, np.int32); t = np.zeros((self.Np, cap)); b = np.zeros((self.Np, cap), np.int8)
        q = np.zeros((self.Np, cap), np.int8); npop = np.zeros((self.Np, n - 1), np.int8)
        self._chk(self.L.pf_get_migrations(self.h, nm.ctypes.data, t.ctypes.data, b.ctypes.data, q.ctypes.data,
                                           npop.ctypes.data, cap))
        return {"n_events": nm, "times": t, "branch": b, "newpop": q, "node_pops": npop}

    def set_timing(self, period):
        self.L.pf_set_timing(self.h, int(period))

    def kernel_times(self):
        """{class: (total_ms_estimate, launches)} from HIP events on the handle's stream."""
        out = {}
        for k, name in enumerate(KERNEL_CLASSES):
            ms = C.c_double(); n = C.c_int64()
            self._chk(self.L.pf_get_kernel_time(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def stats(self):
        a = C.c_int64(); b = C.c_int64(); c = C.c_int64()
        self._chk(self.L.pf_get_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"records": a.value, "state_bytes_per_particle": b.value, "resamples": c.value}

    def set_wg_trace(self, first_step, n_steps):
        """measurement aid (pf_set_wg_trace): time stamps of every workgroup of the row kernel over a range of steps"""
        self._chk(self.L.pf_set_wg_trace(self.h, int(first_step), int(n_steps)))

    def wg_trace(self):
        """[steps, slots, 4] uint64: start, end (10 ns ticks; 0 0 = slot unused), HW_ID | XCC_ID << 32, index in chunk | chunk << 32"""
        info = (C.c_int32 * 2)()
        n = self.L.pf_get_wg_trace(self.h, None, 0, info)
        if n < 0:
            self._chk(-1)
        out = np.zeros(max(n, 0), dtype=np.uint64)
        if n > 0:
            self._chk(0 if self.L.pf_get_wg_trace(self.h, out.ctypes.data_as(C.c_void_p), n, info) >= 0 else -1)
        return out.reshape(info[0], info[1], 4) if n > 0 else out.reshape(0, 0, 4)

    def delay_stats(self):
        """delayed-factor store: factors applied ahead of their position to make room (only with delay_evict; otherwise a full
        store is an error) and the most factors any particle ever had pending"""
        a = C.c_int64(); b = C.c_int32()
        self._chk(self.L.pf_get_delay_stats(self.h, C.byref(a), C.byref(b)))
        return {"forced": a.value, "peak": b.value}


class _PackedModel:
    """Owns the buffers behind a _Model used outside a ParticleFilter (lag calibration)."""

    def __init__(self, m):
        self.ct = np.ascontiguousarray(m["change_times"], dtype=np.float64)
        E = len(self.ct)
        P = int(m.get("n_pops", 1))
        self.ps = np.ascontiguousarray(m["pop_sizes"], dtype=np.float64).reshape(E * P)
        self.rf = np.ascontiguousarray(m.get("record_flags", [3] * E), dtype=np.int32)
        self.lags = np.ascontiguousarray(m.get("lags", np.zeros(E)), dtype=np.float64)
        self.model = _Model(E, P, int(m["nsam"]), 0, float(m["loci_length"]), float(m["mutation_rate"]),
                            float(m["recombination_rate"]), _dp(self.ct), _dp(self.ps), None, None, None,
                            self.rf.ctypes.data_as(C.POINTER(C.c_int32)), _dp(self.lags))
        _attach_structure(self, self.model, m, E, P)


def _pack_model(m):
    pm = _PackedModel(m)
    return pm.model, pm


def median_survival(model, seed=1, min_events=200, max_trees=1000000, device=0):
    """calculate_median_survival_distances (smcsmc.cpp:169-263) on the device; returns (medians[E], trees)."""
    L = load_library()
    mod, keep = _pack_model(model)
    out = np.zeros(mod.n_epochs)
    trees = C.c_int64()
    if L.pf_median_survival(C.byref(mod), int(seed), int(min_events), int(max_trees), out.ctypes.data,
                            C.byref(trees), int(device)) < 0:
        raise PfError(_err(L))
    return out, trees.value


def simulate_sites(model, seed=1, nchunks=1, max_sites=None, device=0):
    """Synthetic data on the device (k_simulate): for each of `nchunks` independent chunks of the model's length the
    ascending site positions and carrier masks.  Returns a list of (positions, masks)."""
    L = load_library()
    L.pf_simulate_sites.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.pf_simulate_sites.restype = C.c_int
    mod, keep = _pack_model(model)
    n = mod.nsam
    if max_sites is None:
        # expected sites 4 N mu H(n-1) L with the largest population size, with ample room
        harmonic = sum(1.0 / i for i in range(1, n))
        nmax = float(np.max(np.asarray(model["pop_sizes"], float)))
        # (populations that stay apart for a long time add the time to their join to every pair across them)
        deep = float(np.max(np.asarray(model["change_times"], float))) if int(model.get("n_pops", 1)) > 1 else 0.0
        max_sites = int(2.0 * (4.0 * nmax * harmonic + n * deep) * model["mutation_rate"] * model["loci_length"]) + 4096
    pos = np.zeros((nchunks, max_sites))
    masks = np.zeros((nchunks, max_sites), np.uint32)
    cnt = np.zeros(nchunks, np.int64)
    if L.pf_simulate_sites(C.byref(mod), int(seed), int(nchunks), int(max_sites), pos.ctypes.data, masks.ctypes.data,
                           cnt.ctypes.data, int(device)) < 0:
        raise PfError(_err(L))
    if (cnt < 0).any():
        raise PfError("pf_simulate_sites: more than max_sites sites in a chunk")
    return [(pos[c, :cnt[c]].copy(), masks[c, :cnt[c]].copy()) for c in range(nchunks)]


def terminal_branch_quantiles(model, seed=1, n_trees=1000000, quantiles=None, device=0):
    """calculate_terminal_branch_length_quantiles (smcsmc.cpp:128-166) on the device:
    returns (lengths[nsam][Q], mean_total_branch_length)."""
    from . import segments as segmod
    L = load_library()
    mod, keep = _pack_model(model)
    q = np.ascontiguousarray(segmod.TBL_QUANTILES if quantiles is None else quantiles, dtype=np.float64)
    out = np.zeros((mod.nsam, len(q)))
    mean = C.c_double()
    if L.pf_terminal_branch_quantiles(C.byref(mod), int(seed), int(n_trees), q.ctypes.data, len(q), out.ctypes.data,
                                      C.byref(mean), int(device)) < 0:
        raise PfError(_err(L))
    return out, mean.value


def calibrated_lags(model, lag_fraction=2.0, seed=1, device=0):
    """CountModel::reset_lag (count.cpp:261-265) with the calibrated survival distances."""
    med, _ = median_survival(model, seed=seed, device=device)
    return med * lag_fraction


# ---- unit-level device entry points (parity tests) ----
def device_math(x, device=0):
    L = load_library()
    x = np.ascontiguousarray(x, dtype=np.float64)
    e = np.zeros_like(x); l = np.zeros_like(x); f = np.zeros_like(x)
    if L.pf_test_math(x.ctypes.data, len(x), e.ctypes.data, l.ctypes.data, f.ctypes.data, device) < 0:
        raise PfError(_err(L))
    return e, l, f


def device_div(a, b, device=0):
    L = load_library()
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    o = np.zeros_like(a)
    if L.pf_test_div(a.ctypes.data, b.ctypes.data, len(a), o.ctypes.data, device) < 0:
        raise PfError(_err(L))
    return o


def device_uniform(seed, slot, stream, first, n, device=0):
    L = load_library()
    o = np.zeros(n)
    if L.pf_test_uniform(seed, slot, stream, first, n, o.ctypes.data, device) < 0:
        raise PfError(_err(L))
    return o


def device_reduce(x, device=0):
    L = load_library()
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = C.c_double(); sc = np.zeros_like(x)
    if L.pf_test_reduce(x.ctypes.data, len(x), C.byref(s), sc.ctypes.data, device) < 0:
        raise PfError(_err(L))
    return s.value, sc


def device_systematic(pilot, device=0):
    """Offspring table of the production resampler for event 0 of seed 1 (u from the resampler stream)."""
    L = load_library()
    x = np.ascontiguousarray(pilot, dtype=np.float64)
    lo = np.zeros(len(x) + 1, np.int32)
    if L.pf_test_systematic(x.ctypes.data, len(x), 0.0, lo.ctypes.data, device) < 0:
        raise PfError(_err(L))
    return lo
