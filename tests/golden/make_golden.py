#!/usr/bin/env python3
"""Generates tests/golden/*.json from the reference's own Python front-end.

Runs ONLY in the build container (needs /root/reference); the outputs are committed so the tests
never touch the reference at run time.  The reference package's __init__ imports tskit (absent),
so the stdlib-only modules that build the binary's command line are imported through a stub
package object.

  cmdlines.json  core_command_line() strings (populationmodels.py:406-437) for several model
                 shapes together with the model tables they encode -- pins the binary's flag parser.
  outfile.json   parse_outfile() (model.py:865-911) applied to .out text -- pins the .out contract.
"""
import json
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref():
    pkg = types.ModuleType("smcsmc")
    pkg.__path__ = [os.path.join(REF, "smcsmc")]
    sys.modules["smcsmc"] = pkg
    import importlib
    pm = importlib.import_module("smcsmc.populationmodels")
    return pm


def cmdlines(pm):
    out = []
    shapes = [
        dict(name="const_1pop_2hap", kw=dict(num_samples=2, sequence_length=1e6)),
        dict(name="const_6epochs", kw=dict(num_samples=4, sequence_length=1e7,
                                            change_points=[0, 0.01, 0.25, 0.5, 1, 1.5],
                                            population_sizes=[[1], [1], [1], [1], [1], [1]])),
        dict(name="bottleneck", kw=dict(num_samples=4, sequence_length=1e6, N0=14312, mutation_rate=1.25e-8,
                                        recombination_rate=3.5e-9, change_points=[0, 0.1, 0.5],
                                        population_sizes=[[1.5], [0.3], [2.0]])),
        dict(name="two_pop_im", kw=dict(num_samples=8, sequence_length=1e6, num_populations=2,
                                        sample_populations=[1, 1, 1, 1, 2, 2, 2, 2],
                                        change_points=[0, 0.1, 0.5],
                                        population_sizes=[[1, 1], [2, 2], [2, 2]],
                                        migration_rates=[[[0, 1], [1, 0]], [[0, 1], [1, 0]], [[0, 0], [0, 0]]],
                                        migration_commands=[None, None, "-ej 0.5 2 1"])),
    ]
    for sh in shapes:
        pop = pm.Population(**sh["kw"])
        for vb in (False, True):
            line = pop.core_command_line(vb=vb)
            out.append(dict(name=sh["name"], vb=vb, cmdline=line, N0=pop.N0, mutation_rate=pop.mutation_rate,
                            recombination_rate=pop.recombination_rate, sequence_length=pop.sequence_length,
                            num_samples=pop.num_samples, num_populations=pop.num_populations,
                            change_points=list(pop.change_points), population_sizes=pop.population_sizes,
                            migration_rates=pop.migration_rates))
    return out


def outfile():
    import importlib
    model = importlib.import_module("smcsmc.model")
    src = os.path.join(HERE, "sample.out")
    s = model.Smcsmc.__new__(model.Smcsmc)
    data = model.Smcsmc.parse_outfile(s, src)
    rows = []
    for (key, label), val in sorted(data.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
        rows.append(dict(type=key[0], epoch=key[1], frm=key[2], to=key[3], clump=key[4], label=label, value=val))
    return rows


def mstep(pm):
    """Smcsmc.m_step (model.py:989-1048) applied to hand-made sufficient statistics: plain and variational-Bayes
    variants, one- and two-population models, with and without -maxNE / -no_infer_recomb."""
    import importlib
    import random
    model = importlib.import_module("smcsmc.model")
    cases = []
    rnd = random.Random(12345)
    for name, P, E, vb, maxne, infer in [("one_pop", 1, 5, False, 1e99, True), ("one_pop_vb_cap", 1, 4, True, 30000.0, True),
                                         ("two_pop", 2, 3, False, 1e99, True), ("two_pop_vb_norecomb", 2, 3, True, 1e99, False)]:
        pop = pm.Population(num_samples=4, sequence_length=1e6, num_populations=P,
                            sample_populations=[1 + (i * P) // 4 for i in range(4)],
                            change_points=[0] + [0.1 * (k + 1) for k in range(E - 1)],
                            population_sizes=[[1.0] * P for _ in range(E)],
                            migration_rates=[[[0.0 if a == b else 0.5 for b in range(P)] for a in range(P)] for _ in range(E)])
        pop._finalize_and_validate()      # fills the default event counts the front-end carries along
        s = model.Smcsmc.__new__(model.Smcsmc)
        s.pop = pop; s.do_m_step = True; s.vb = vb; s.maxNE = maxne; s.infer_recomb = infer
        s.vb_dirichlet = {"ne": [1.0, 1.0], "migr": [1.0, 1.0]}
        data = {}
        rows = []
        for e in range(E):
            for a in range(P):
                opp, cnt = rnd.uniform(1e5, 1e7), rnd.uniform(1.0, 400.0)
                data[(("Coal", e, a, -1, -1), "Opp")] = opp; data[(("Coal", e, a, -1, -1), "Count")] = cnt
                rows.append(dict(type="Coal", epoch=e, frm=a, to=-1, opp=opp, count=cnt))
                for b in range(P):
                    if a != b:
                        opp, cnt = rnd.uniform(1e5, 1e7), rnd.uniform(0.0, 50.0)
                        data[(("Migr", e, a, b, -1), "Opp")] = opp; data[(("Migr", e, a, b, -1), "Count")] = cnt
                        rows.append(dict(type="Migr", epoch=e, frm=a, to=b, opp=opp, count=cnt))
        opp, cnt = rnd.uniform(1e10, 1e11), rnd.uniform(100.0, 2000.0)
        data[(("Recomb", -1, -1, -1, -1), "Opp")] = opp; data[(("Recomb", -1, -1, -1, -1), "Count")] = cnt
        rows.append(dict(type="Recomb", epoch=-1, frm=-1, to=-1, opp=opp, count=cnt))
        rho0 = pop.recombination_rate
        s.m_step(data)
        cases.append(dict(name=name, P=P, E=E, vb=vb, maxNE=maxne, infer_recomb=infer, N0=pop.N0, rows=rows,
                          recombination_rate_before=rho0, population_sizes=pop.population_sizes,
                          migration_rates=pop.migration_rates, recombination_rate=pop.recombination_rate,
                          next_cmdline=pop.core_command_line(vb=vb)))
    return cases


def seg_prefix(src_rel, dst_name, limit):
    """First `limit` bp of one of the reference's committed scrm data sets (a data fixture of its own tests),
    closed with an all-missing row at the cut like convert_scrm_to_seg does (populationmodels.py:535-575)."""
    src = os.path.join(REF, src_rel)
    out = []
    nsam = None
    for line in open(src):
        f = line.rstrip("\n").split("\t")
        start, length = int(f[0]), int(f[1])
        nsam = len(f[-1])
        if start + length > limit:
            break
        out.append(line)
    end = int(out[-1].split("\t")[0]) + int(out[-1].split("\t")[1])
    if end < limit:
        out.append("%d\t%d\tT\tF\t1\t%s\n" % (end, limit - end, "." * nsam))
    open(os.path.join(HERE, "seg", dst_name), "w").write("".join(out))


if __name__ == "__main__":
    pm = load_ref()
    seg_prefix("test/old/newtests/testdata/twopopssplit_unidirmigr.seg", "twopopssplit_unidirmigr_first2Mb.seg", 2000001)
    # the whole 10 Mb data set of the reference's constant-size regression test (test_const_pop_size.py:15-49)
    import shutil
    shutil.copyfile(os.path.join(REF, "test/old/newtests/testdata/constpopsize.seg"), os.path.join(HERE, "seg", "constpopsize.seg"))
    json.dump(cmdlines(pm), open(os.path.join(HERE, "cmdlines.json"), "w"), indent=1)
    json.dump(mstep(pm), open(os.path.join(HERE, "mstep.json"), "w"), indent=1)
    if os.path.exists(os.path.join(HERE, "sample.out")):
        json.dump(outfile(), open(os.path.join(HERE, "outfile.json"), "w"), indent=1)
    print("golden fixtures written")
