#!/usr/bin/env python3
"""Generates tests/golden/scrm_to_seg.json: what the reference's own scrm -> .seg converter
(smcsmc/populationmodels.py:502-577, Population.convert_scrm_to_seg) writes for a few hand-written scrm outputs.

Runs ONLY in the build container (needs /root/reference; the function is imported through a stub package, as
make_reference_bands.py does).  The inputs below are written by hand in scrm's output format (a `positions:` line with
positions in [0, 1), then one 0/1 string per haplotype); the fixture stores them together with the rows the reference
writes, so that tests/test_host_cpu.py can hold smcsmc_amd.simulate.sites_to_seg / write_seg -- the .seg conventions of this
build's data simulators -- to them: positions int(x * L + 0.5), a leading position 1, a final all-missing row up to L,
missing leaves as '.'.  (The unphased branch of the reference's function divides a list length with `/` and does not run
under Python 3; only the phased form is pinned.)
"""
import importlib
import json
import os
import sys
import tempfile
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    dict(name="four_haplotypes", L=1000, n=4, missing=[],
         positions=[0.0104, 0.0731, 0.2500, 0.4127, 0.4135, 0.7777, 0.9301],
         haplotypes=["0100101", "1100001", "0011100", "0010110"]),
    dict(name="rounding_half_up", L=200, n=2, missing=[],
         positions=[0.0125, 0.0475, 0.5025, 0.9949],            # x * L = 2.5, 9.5, 100.5, 198.98
         haplotypes=["0110", "1011"]),
    dict(name="missing_leaves", L=5000, n=6, missing=[1, 4],
         positions=[0.00031, 0.1, 0.10002, 0.5, 0.99999],
         haplotypes=["01001", "11111", "00100", "10010", "00000", "01110"]),
    dict(name="single_site", L=100, n=2, missing=[], positions=[0.5], haplotypes=["0", "1"]),
]


def main():
    pkg = types.ModuleType("smcsmc")
    pkg.__path__ = [os.path.join(REF, "smcsmc")]
    sys.modules["smcsmc"] = pkg
    pm = importlib.import_module("smcsmc.populationmodels")
    out = []
    tmp = tempfile.mkdtemp(prefix="scrm2seg_")
    for c in CASES:
        pop = pm.Population(sequence_length=c["L"], num_samples=c["n"], scrmpath="scrm")
        infile = os.path.join(tmp, c["name"] + ".scrm")
        with open(infile, "w") as f:
            f.write("scrm %d 1 -t 10 -r 4 %d\n1 2 3\n\n//\nsegsites: %d\n" % (c["n"], c["L"], len(c["positions"])))
            f.write("positions: " + " ".join(repr(p) for p in c["positions"]) + "\n")
            for hpl in c["haplotypes"]:
                f.write(hpl + "\n")
        outfile = os.path.join(tmp, c["name"] + ".seg")
        pop.convert_scrm_to_seg(infile, outfile, c["missing"], True)
        rows = [ln.rstrip("\n").split("\t") for ln in open(outfile)]
        out.append(dict(c, seg_rows=rows))
        print(c["name"], len(rows), "rows")
    json.dump(dict(generator="tests/golden/make_scrm_to_seg.py", source="smcsmc/populationmodels.py:502-577", cases=out),
              open(os.path.join(HERE, "scrm_to_seg.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
