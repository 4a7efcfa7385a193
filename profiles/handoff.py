#!/usr/bin/env python3
"""The cost of handing a row over between the extend wavefronts: a kernel boundary per row against a resident grid (GPU box).

    python profiles/handoff.py [--nw 157] [--rows 4000]

Prints microseconds per row for 0, 5, 10 and 20 us of stand-in work per wavefront and row; the difference between the two modes
at equal work is what a resident row kernel could save on a row that does not resample."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smcsmc_amd import pf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nw", type=int, default=157)
    ap.add_argument("--rows", type=int, default=4000)
    args = ap.parse_args()
    print("| work per wavefront and row | launch per row (us/row) | resident grid, every wavefront arrives and polls | resident grid, one arrival and poller per workgroup |")
    print("|---|---|---|---|")
    for spin in (0.0, 5.0, 10.0, 20.0):
        res = [min(pf.probe_handoff(mode, args.rows, args.nw, spin)[0] for _ in range(3)) for mode in (0, 1, 2)]
        print("| %.0f us | %.2f | %.2f | %.2f |" % (spin, res[0], res[1], res[2]), flush=True)
    cs = [pf.probe_handoff(mode, 500, args.nw, 0.0)[1] for mode in (0, 1, 2)]
    print("checksums of the three forms after 500 rows (the same numbers must come out): %r %r %r -> %s" % (cs[0], cs[1], cs[2], "equal" if cs[0] == cs[1] == cs[2] else "DIFFERENT"))
    for nw in (40, 157, 628):
        res = [min(pf.probe_handoff(mode, args.rows, nw, 0.0)[0] for _ in range(3)) for mode in (0, 1, 2)]
        print("| no work, %d wavefronts | %.2f | %.2f | %.2f |" % (nw, res[0], res[1], res[2]), flush=True)


if __name__ == "__main__":
    main()
