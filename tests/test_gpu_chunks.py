"""Several chunks in one process (bin/smcsmc -chunks K -ranks R): the native counterpart of the front-end's process farm
and file sum (smcsmc/model.py:1050-1100, 1176-1184).  The statistics are exchanged once per E-step (RCCL all-gather when
every rank has a device of its own, host memory otherwise) and summed in chunk order, so the result may not depend on how
many ranks shared the work."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "smcsmc")
SEG = os.path.join(ROOT, "tests", "golden", "seg", "constpopsize.seg")
L = 4000000
CORE = ("-N0 10000 -t %g -r %g %d -eN 0 1 -eN 0.01 1 -eN 0.25 1 -eN 1 1" % (4e4 * 2.5e-8 * L, 4e4 * 1e-8 * L, L)).split()
COMMON = ["-nsam", "2", "-seg", SEG, "-Np", "300", "-tmax", "4", "-lag", "30000", "-seed", "5"]


def _run(tmp_path, name, extra):
    r = subprocess.run([BIN] + CORE + COMMON + extra + ["-o", str(tmp_path / name)], capture_output=True, text=True)
    assert r.returncode == 0, (name, r.stderr[-400:])
    return open(tmp_path / (name + ".out")).read(), r.stderr


def _rows(text):
    rows = [ln.split() for ln in text.splitlines()[1:]]
    return {(r[4], int(r[1]), int(r[5]), int(r[6])): (float(r[7]), float(r[8])) for r in rows if r[0] == rows[-1][0]}


def test_chunk_statistics_do_not_depend_on_the_ranks(hiplib, tmp_path):
    one, log1 = _run(tmp_path, "r1", ["-chunks", "4", "-ranks", "1"])
    assert "exchanged by rccl" in log1                      # one rank, one device: the RCCL all-gather runs (world 1)
    two, log2 = _run(tmp_path, "r2", ["-chunks", "4", "-ranks", "2", "-reduce", "host"])
    four, _ = _run(tmp_path, "r4", ["-chunks", "4", "-ranks", "4", "-reduce", "host"])
    assert "2 rank(s)" in log2
    assert one == two == four                               # bit-identical .out files
    d = _rows(one)
    # four chunks bring four sets of prior pseudo-counts (count.cpp:161-227), as four .out files added up would
    assert d[("LogL", -1, -1, -1)][0] == 1.0 and d[("Coal", 3, 0, -1)][0] > 4.0
    # and the sum is that of the chunks filtered one at a time with the same seeds (same data window, same per-bp rates)
    single = _rows(_run(tmp_path, "whole", [])[0])
    assert abs(d[("LogL", -1, -1, -1)][1] / single[("LogL", -1, -1, -1)][1] - 1) < 0.02     # chunk starts from fresh priors cost little
    assert abs(d[("Recomb", -1, -1, -1)][1] / single[("Recomb", -1, -1, -1)][1] - 1) < 0.1


def test_six_chunks_on_a_device_take_the_narrow_count_columns(hiplib, tmp_path):
    """With six or more chunks per device the host sets pf_params.count_wgs (one per 256 particles: the library then tapers the columns of
    the young epochs; the sums are grouped by workgroup): the same bytes as asking for that with -count_wgs, and the same statistics to
    rounding as other widths give.  A width pinned with -count_wgs gives the same bytes for any number of ranks."""
    auto, _ = _run(tmp_path, "auto", ["-chunks", "6", "-ranks", "1"])
    pinned, _ = _run(tmp_path, "pin", ["-chunks", "6", "-ranks", "1", "-count_wgs", "2"])
    assert auto == pinned
    spread, _ = _run(tmp_path, "spread", ["-chunks", "6", "-ranks", "3", "-reduce", "host", "-count_wgs", "2"])
    assert spread == pinned
    wide = _rows(_run(tmp_path, "wide", ["-chunks", "6", "-ranks", "1", "-count_wgs", "1"])[0])
    for key, (count, opp) in _rows(auto).items():
        assert count == pytest.approx(wide[key][0], rel=1e-6, abs=1e-9) and opp == pytest.approx(wide[key][1], rel=1e-6, abs=1e-9), key


def test_chunks_leave_their_local_recombination_maps(hiplib, tmp_path):
    """The reference writes <prefix>.recomb.gz in every chunk process (smcsmc.cpp:376-383); -chunks K leaves one map per chunk,
    <prefix>.chunkN.recomb.gz, with the chunk's own loci, whichever way the chunks are spread over ranks."""
    import gzip
    _run(tmp_path, "m", ["-chunks", "2", "-ranks", "1"])
    rows = []
    for c in range(2):
        path = tmp_path / ("m.chunk%d.recomb.gz" % c)
        assert path.exists(), "no local recombination map for chunk %d" % c
        lines = gzip.open(path, "rt").read().splitlines()
        assert lines[0].split()[:4] == ["iter", "locus", "size", "opp_per_nt"]
        body = [ln.split() for ln in lines[1:]]
        assert len(body) == L // 2 // 100
        assert float(body[0][1]) == 1 + c * (L // 2)                 # first locus of the chunk (1-based start position)
        assert sum(float(r[3]) for r in body) > 0
        rows.append(body)
    # the same chunks filtered one after the other by two ranks (no lockstep launch) leave the same maps
    _run(tmp_path, "m2", ["-chunks", "2", "-ranks", "2", "-reduce", "host"])
    for c in range(2):
        other = [ln.split() for ln in gzip.open(tmp_path / ("m2.chunk%d.recomb.gz" % c), "rt").read().splitlines()[1:]]
        a = np.array([[float(v) for v in r_[3:]] for r_ in other]); b = np.array([[float(v) for v in r_[3:]] for r_ in rows[c]])
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-12)      # (five significant digits in the file; sums of atomics)


def test_rccl_exchange_between_two_devices(hiplib, tmp_path):
    """-reduce rccl with two ranks on two devices: the branch no one-GPU box can run.  Switches itself on where a second device
    is visible (the driver's multi-GPU node) and must then give the .out of the host exchange, byte for byte."""
    from smcsmc_amd import pf
    if pf.load_library().pf_device_count() < 2:
        pytest.skip("one device: RCCL between ranks needs two")
    host, _ = _run(tmp_path, "h2", ["-chunks", "4", "-ranks", "2", "-reduce", "host"])
    rccl, log = _run(tmp_path, "n2", ["-chunks", "4", "-ranks", "2", "-devices", "2", "-reduce", "rccl"])
    assert "exchanged by rccl" in log and "2 device(s)" in log
    assert rccl == host


def test_chunked_em_iterations(hiplib, tmp_path):
    """-EM with several chunks: the M-step works on the summed statistics, every iteration re-filters all chunks."""
    text, _ = _run(tmp_path, "em", ["-chunks", "2", "-ranks", "2", "-reduce", "host", "-EM", "1"])
    iters = sorted({int(ln.split()[0]) for ln in text.splitlines()[1:]})
    assert iters == [0, 1]
    ll = [float(ln.split()[8]) for ln in text.splitlines()[1:] if ln.split()[4] == "LogL"]
    assert len(ll) == 2 and all(np.isfinite(ll))


def test_bench_two_ranks_rehearsal(hiplib):
    """bench.py's multi-rank flow (one process per rank under torch.distributed.run, barrier + MAX of the times, ordered
    sum of the gathered statistics, one JSON line from rank 0) on this one-GPU box: both ranks on device 0, collectives
    through gloo (--rehearse-on-one-gpu; RCCL itself is exercised by bin/smcsmc -chunks above).  The aggregate must count
    the segments of both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "0", "--length", "2e6", "--particles", "1000", "--no-cpu"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, cwd=root, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu"] + common,
                         capture_output=True, text=True, cwd=root, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["metric"] == j1["metric"]
    seg1 = j1["config"]["segments_per_chunk"]
    assert j2["value"] * j2["ms_per_step"] * 1e-3 > 1.5 * seg1            # both ranks' segments are in the aggregate
    assert np.isfinite(j2["config"]["log_likelihood_sum"]) and j2["config"]["log_likelihood_sum"] < j1["config"]["log_likelihood_sum"]
