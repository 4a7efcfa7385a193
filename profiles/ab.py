#!/usr/bin/env python3
"""A/B of bench.py variants on one box, alternating (two boxes differ by up to 5 %):

    python profiles/ab.py --variants "0;2048;--debug 0 --count-wgs 160" --rounds 2 -- --length 2e7 --steps 3 --warmup 1 --no-cpu

A variant is either a number (pf_params.debug) or a string of extra bench.py flags.  Prints value, ms per step, the row kernel's
average launch and the log-likelihood per run; what follows `--` goes to every run."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    rest = [a for a in args.rest if a != "--"]
    variants = [v.strip() for v in args.variants.split(";")]
    for rnd in range(args.rounds):
        for v in variants:
            extra = ["--debug", v] if v.lstrip("-").isdigit() else v.split()
            tree = ROOT
            if extra and extra[0].startswith("@"):            # "@scratch/r3tree ..." = the bench.py (and library) of another checkout
                tree = os.path.join(ROOT, extra[0][1:]); extra = extra[1:]
            r = subprocess.run([sys.executable, os.path.join(tree, "bench.py")] + rest + extra, capture_output=True, text=True, cwd=tree)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print("%-40s FAILED %s" % (v, r.stderr[-300:].replace("\n", " | ")), flush=True)
                continue
            d = json.loads(line[-1])
            rf = d.get("roofline") or {}
            print("%-40s %10.1f %s  %9.2f ms/step  launch %.2f us  logl %s" % (v, d["value"], d["unit"], d["ms_per_step"], rf.get("avg_launch_us") or float("nan"),
                                                                               d["config"].get("log_likelihood")), flush=True)


if __name__ == "__main__":
    main()
