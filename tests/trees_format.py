"""The line format of <prefix>.trees.gz, read off the reference's own example (tests/golden/ex.trees.gz = the reference's
test/data/ex.trees.gz, the input of its trees2tskit test): six tab-separated columns -- event code R / C / M, position and
height with one decimal, from and to population, the descendants as a bit string that starts with sample 1
(ParticleContainer::printTrees, pc.cpp:515-555; print_descendants, descendants.hpp:51-65)."""
import gzip
import os
import re

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LINE = re.compile(r"^([RCM])\t(\d+\.\d)\t(\d+\.\d)\t(-?\d+)\t(-?\d+)\t(0|[01]*1)$")


def reference_example():
    return gzip.open(os.path.join(GOLD, "ex.trees.gz"), "rt").read().splitlines()


def check_lines(lines, nsam, npop):
    """Every line follows the grammar of the reference's example, and the columns mean what they mean there."""
    last_pos = None
    for ln in lines:
        m = LINE.match(ln)
        assert m, "not a .trees.gz line: %r" % ln
        code, pos, hgt, frm, to, desc = m.group(1), float(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6)
        assert len(desc) <= nsam and hgt >= 0.0
        if code == "R":
            assert frm == -1 and to == -1
        elif code == "C":
            assert 0 <= frm < npop and to == -1
        else:
            assert 0 <= frm < npop and 0 <= to < npop and frm != to
        assert last_pos is None or pos <= last_pos, "events are written last position first"
        last_pos = pos
