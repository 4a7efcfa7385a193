"""EM outer loop on the GPU: the binary's own -EM iterations (CountModel::reset_model_parameters,
count.cpp:44-63, 267-352) and the in-process loop of smcsmc_amd.em (model.py:989-1184)."""
import os
import subprocess

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "smcsmc")
SEG = os.path.join(ROOT, "tests", "golden", "seg", "constpopsize_first3000.seg")


def _rows(path):
    from smcsmc_amd import outfile
    lines = open(path).read().splitlines()
    by_iter = {}
    for ln in lines[1:]:
        by_iter.setdefault(int(ln.split()[0]), []).append(ln)
    parsed = {it: outfile.parse_outfile("\n".join([lines[0]] + rows), is_text=True) for it, rows in by_iter.items()}
    return by_iter, parsed


def test_binary_em_iterations(hiplib, tmp_path):
    if not os.path.exists(BIN):
        from smcsmc_amd import build
        build.build_all()
    L = 2000000
    # start from a model that is wrong by a factor 3 in every epoch (truth: N = 1e4 everywhere, rho = 1e-8)
    core = ("-N0 10000 -t %g -r %g %d -eN 0 3 -eN 0.05 3 -eN 0.25 3 -eN 1 3"
            % (4 * 1e4 * 2.5e-8 * L, 4 * 1e4 * 1e-8 * L, L)).split()
    common = core + ["-nsam", "2", "-Np", "500", "-tmax", "4", "-lag", "20000", "-seed", "7", "-seg", SEG]
    r = subprocess.run([BIN] + common + ["-EM", "2", "-o", str(tmp_path / "em")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r0 = subprocess.run([BIN] + common + ["-EM", "0", "-o", str(tmp_path / "e0")], capture_output=True, text=True)
    assert r0.returncode == 0, r0.stderr
    by_iter, parsed = _rows(tmp_path / "em.out")
    assert sorted(by_iter) == [0, 1, 2]
    # the first E-step is the plain run
    assert by_iter[0] == open(tmp_path / "e0.out").read().splitlines()[1:]
    # the M-step is reported, and the second E-step ran under the updated model: its pseudo-counts stay those of the
    # initial model (CountModel is constructed once, smcsmc.cpp:77), its likelihood improves
    assert "MODEL IS RESET" in r.stderr and "Setting size of population 0" in r.stderr
    ll = [parsed[i][(("LogL", -1, -1, -1, -1), "Count")] for i in range(3)]
    assert ll[1] > ll[0] and ll[2] > ll[0]
    ne = lambda d, e: d[(("Coal", e, 0, -1, -1), "Opp")] / (2 * d[(("Coal", e, 0, -1, -1), "Count")])     # noqa: E731
    # estimates move from the wrong 30000 towards the truth 10000 in the well-informed epochs
    for e in (1, 2):
        assert abs(ne(parsed[2], e) - 1e4) < abs(3e4 - 1e4)
        assert ne(parsed[2], e) < 2.2e4


def test_in_process_em_loop(hiplib):
    from smcsmc_amd import em, segments as segmod
    n, L = 4, 1.5e6
    truth = cases.make_model(n=n, E=6, L=L)
    chunks = []
    for seed in (11, 12):
        packed = cases.make_segments(truth, seed=seed, max_seg_len=5000)
        S = segmod.Segments.from_pieces(packed["start"].astype(np.int64) + 1, packed["length"].astype(np.int64), packed["state"],
                                        packed["alleles"], n, L)
        chunks.append(S)
    cp = list(np.array(truth["change_times"]) / 4e4)
    pop = em.PopulationModel(N0=10000, sequence_length=L, num_samples=n, change_points=cp,
                             population_sizes=[[2.5]] * 6)                    # 2.5x too large everywhere
    seen = []
    final, hist = em.run_em(pop, chunks, iterations=2, np_particles=400, seed=3,
                            on_iteration=lambda it, p, d: seen.append((it, [row[0] for row in p.population_sizes])))
    assert len(hist) == 3 and [s[0] for s in seen] == [0, 1, 2]
    ll = [h[(("LogL", -1, -1, -1, -1), "Count")] for h in hist]
    assert ll[2] > ll[0]
    sizes = np.array([row[0] for row in final.population_sizes])
    assert (sizes[1:5] < 2.0).all() and (sizes[1:5] > 0.5).all()             # moved towards 1.0
    # the argv a front-end would launch next is well-formed for the binary's parser
    line = final.core_command_line()
    assert line.startswith("-N0 10000 -t ") and line.count("-eN") == 6


def test_guided_em_loop(hiplib):
    """alpha > 0: the local recombination map of iteration i becomes the recombination guide of iteration i + 1
    (Smcsmc.do_iteration, model.py:1129-1143; processrecombination.py) without leaving the process."""
    from smcsmc_amd import em, segments as segmod
    n, L = 4, 1.5e6
    truth = cases.make_model(n=n, E=6, L=L)
    chunks = []
    for seed in (21, 22):
        packed = cases.make_segments(truth, seed=seed, max_seg_len=5000)
        S = segmod.Segments.from_pieces(packed["start"].astype(np.int64) + 1, packed["length"].astype(np.int64), packed["state"],
                                        packed["alleles"], n, L)
        chunks.append(S)
    cp = list(np.array(truth["change_times"]) / 4e4)
    pop = em.PopulationModel(N0=10000, sequence_length=L, num_samples=n, change_points=cp, population_sizes=[[1.0]] * 6)
    guides = {}
    final, hist = em.run_em(pop, chunks, iterations=2, np_particles=2000, seed=5, alpha=0.5, beta=4.0, guides=guides)
    assert sorted(guides) == [0, 1]
    for g in guides.values():
        assert g["positions"][0] == 0 and (np.diff(g["positions"]) > 0).all() and g["positions"][-1] < L
        assert (g["rates"] > 0).all() and (g["leaf_rates"] > 0).all()
        assert g["leaf_rates"].shape == (len(g["positions"]), n)
        assert abs(np.average(g["rates"], weights=np.diff(np.append(g["positions"], L))) / 1e-8 - 1) < 0.5
    ll = [h[(("LogL", -1, -1, -1, -1), "Count")] for h in hist]
    assert np.isfinite(ll).all() and abs(ll[2] - ll[0]) < 0.02 * abs(ll[0])
    # guided sampling is an importance sampler of the same posterior: the estimates stay where the unguided ones are
    rho = [h[(("Recomb", -1, -1, -1, -1), "Count")] / h[(("Recomb", -1, -1, -1, -1), "Opp")] for h in hist]
    assert abs(rho[2] / rho[0] - 1) < 0.15
    sizes = np.array([row[0] for row in final.population_sizes])
    assert (sizes[2:5] > 0.7).all() and (sizes[2:5] < 1.4).all()          # epochs 0 and 1 hold a handful of events
