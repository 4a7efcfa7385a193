#!/usr/bin/env python3
"""Generates tests/golden/reference_bands.json: the acceptance bands of the reference's own regression tests.

Runs ONLY in the build container (needs /root/reference).  The reference's stale statistical tests
(test/old/newtests/test_const_pop_size.py, test_two_pops.py) hold the only numbers the reference
itself states for the hot path: ranges of the Ne / recombination / migration estimates after running the filter on
committed scrm data with fixed flags.  This script instantiates those unittest classes (their setUp() only fills in
attributes), reads `targets` and lets the reference's own harness build the inference command line
(TestGeneric.build_command, test_generic.py:105-224), so neither the bands nor the flags are transcribed by hand.

The front-end the old tests call (`../python/smcsmc.py`) passes everything it does not consume to the binary
(model.py:1057-1092); `binary_argv` below is the command with the front-end-only tokens removed (-EM is forced to 0 per
E-step by the front-end, -no_m_step / -alpha / -chunks are front-end options).

The data files the classes name are the reference's committed fixtures; they are data and are kept under
tests/golden/seg/ (30 Mb ones gzip-compressed).
"""
import gzip
import importlib
import json
import os
import shutil
import sys
import types

REF = "/root/reference"
NEWTESTS = os.path.join(REF, "test", "old", "newtests")
HERE = os.path.dirname(os.path.abspath(__file__))

CLASSES = [
    ("test_const_pop_size", "TestConstPopSize", "test_const_pop_size.py:13-50"),
    ("test_const_pop_size", "TestConstPopSize_MissingData", "test_const_pop_size.py:150-172"),
    ("test_const_pop_size", "TestConstPopSize_FourEpochs", "test_const_pop_size.py:113-145"),
    ("test_const_pop_size", "TestConstPopSize_FourEpochs_MissingData", "test_const_pop_size.py:177-199"),
    ("test_const_pop_size", "TestConstPopSize_FourEpochs_EightSamples", "test_const_pop_size.py:202-229"),
    ("test_const_pop_size", "TestConstPopSize_FourEpochs_FalseStart", "test_const_pop_size.py:232-246"),
    ("test_const_pop_size", "TestConstPopSize_Migration", "test_const_pop_size.py:249-318"),
    ("test_two_pops", "TestTwoPopsSplitUniDirMigr", "test_two_pops.py:51-119"),
    # the remaining classes with committed data: the no-data focused-sampling checks (the reference's direct test of
    # biased tree-point sampling with delayed importance weights, with min-ESS targets) ...
    ("test_bias_nodata", "TestBias_00_NoBias", "test_bias_nodata.py:15-49"),
    ("test_bias_nodata", "TestBias_01_Bias2", "test_bias_nodata.py:53-63"),
    ("test_bias_nodata", "TestBias_02_Bias5", "test_bias_nodata.py:67-78"),
    ("test_bias_nodata", "TestBias_0_Migr_NoBias", "test_bias_nodata.py:81-122"),
    ("test_bias_nodata", "TestBias_1_Migr_Bias2", "test_bias_nodata.py:126-136"),
    # ... and the other two-population split scenarios (30 Mb, 5 E-steps each)
    ("test_two_pops", "TestTwoPopsSplitUniDirMigrInRecentEpoch", "test_two_pops.py:234-301"),
    ("test_two_pops", "TestTwoPopsSplitUniDirMigrInMidEpoch", "test_two_pops.py:303-371"),
    ("test_two_pops", "TestTwoPopsSplitUniDirMigr_bs3", "test_two_pops.py:376-424"),
    ("test_two_pops", "TestTwoPopsSplitUniDirMigrInRecentEpoch_bs3", "test_two_pops.py:429-477"),
    ("test_two_pops", "TestTwoPopsSplitUniDirMigrInMidEpoch_bs3", "test_two_pops.py:481-529"),
]


def load_modules():
    pkg = types.ModuleType("smcsmc")
    pkg.__path__ = [os.path.join(REF, "smcsmc")]
    sys.modules["smcsmc"] = pkg
    ctx = types.ModuleType("context")
    ctx.populationmodels = importlib.import_module("smcsmc.populationmodels")
    ctx.execute = importlib.import_module("smcsmc.execute")
    sys.modules["context"] = ctx
    sys.path.insert(0, NEWTESTS)
    return {m: importlib.import_module(m) for m in ("test_const_pop_size", "test_two_pops", "test_bias_nodata")}


def binary_argv(cmd):
    """The tokens the binary sees: drop the front-end path and front-end-only options."""
    toks = cmd.split()[1:]
    out = []
    i = 0
    while i < len(toks):
        t = toks[i]
        if t in ("-EM", "-alpha", "-chunks"):
            i += 2
            continue
        if t in ("-no_m_step", "-no_infer_recomb"):
            i += 1
            continue
        if t == "-seg":
            out += ["-seg", "@SEG@"]
            i += 2
            continue
        out.append(t)
        i += 1
    return out


def main():
    mods = load_modules()
    cases = []
    os.makedirs(os.path.join(HERE, "seg"), exist_ok=True)
    for modname, clsname, where in CLASSES:
        cls = getattr(mods[modname], clsname)
        t = cls("test_inference")
        t.setUp()
        t.pop.filename = t.prefix + t.filename_disambiguator + ".seg"
        cmd = t.build_command()
        data = os.path.basename(t.pop.filename)
        src = os.path.join(NEWTESTS, "testdata", data)
        assert os.path.exists(src), src
        big = os.path.getsize(src) > 1 << 20
        dst = os.path.join(HERE, "seg", data + (".gz" if big else ""))
        if not os.path.exists(dst):
            if big:
                with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
                    shutil.copyfileobj(f, g)
            else:
                shutil.copyfile(src, dst)
        targets = []
        for tg in t.targets:
            d = {k: tg[k] for k in tg if k in ("type", "pop", "epoch", "from_pop", "to_pop", "min", "max", "truth", "ess")}
            targets.append(d)
        cases.append(dict(
            name=clsname, source="test/old/newtests/" + where, data=os.path.basename(dst),
            front_end_command=cmd, binary_argv=binary_argv(cmd), em_iterations=t.em, np=t.np, seed=list(t.seed),
            nsam=t.pop.num_samples, sequence_length=t.pop.sequence_length, missing_leaves=list(t.missing_leaves),
            max_out_of_range=t.max_out_of_range, targets=targets))
    json.dump(dict(generator="tests/golden/make_reference_bands.py", cases=cases),
              open(os.path.join(HERE, "reference_bands.json"), "w"), indent=1)
    for c in cases:
        print(c["name"], c["data"], len(c["targets"]), "targets")
        print("   ", " ".join(c["binary_argv"]))


if __name__ == "__main__":
    main()
