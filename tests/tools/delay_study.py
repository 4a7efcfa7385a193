#!/usr/bin/env python3
"""Focused sampling with data: does the offset of the two-population classes against the reference's bands shrink with the number
of particles (a finite-Np effect of delaying the weights) or stay (a difference of schedule)?  And how full does the store of
delayed factors get?  (GPU box.)

    python tests/tools/delay_study.py --classes TestTwoPopsSplitUniDirMigr --np 1000,4000,16000 --delay 0.5,0.25 --seeds 2

Every run goes through bin/smcsmc with the class's own command line plus -Np / -delay; the log line of the delayed-factor store
(peak pending per particle, factors applied early) is read from the binary's stderr.  Writes <out>.json / <out>.md.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import reference_bands as rb  # noqa: E402


def run(case, seed, tmpdir, extra):
    seg = rb.seg_path(case, tmpdir)
    prefix = os.path.join(tmpdir, "%s_s%d" % (case["name"], seed))
    r = subprocess.run(rb.argv_for(case, seed, seg, prefix, extra), capture_output=True, text=True, timeout=3000)
    if r.returncode != 0:
        return None, r.stderr[-300:]
    est = rb.read_estimates(prefix + ".out")
    for suffix in (".out", ".log", ".recomb.gz"):
        try: os.unlink(prefix + suffix)
        except OSError: pass
    peaks = [int(x) for x in re.findall(r"at most (\d+) pending", r.stderr)]
    forced = [int(x) for x in re.findall(r"; (\d+) applied early", r.stderr)]
    return est, dict(peak=max(peaks) if peaks else 0, forced=sum(forced))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", default="TestTwoPopsSplitUniDirMigr")
    ap.add_argument("--np", default="1000,4000,16000")
    ap.add_argument("--delay", default="0.5,0.25")
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--extra", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "delay_study"))
    args = ap.parse_args()
    cases = {c["name"]: c for c in rb.load_cases(variants=False)}
    tmpdir = tempfile.mkdtemp(prefix="delaystudy_")
    rows = []
    md = ["| class | Np | -delay | target | band | " + " | ".join("seed %d" % s for s in range(1, args.seeds + 1)) + " | store: peak pending / applied early |", "|---|---|---|---|---|" + "---|" * (args.seeds + 1)]
    for name in args.classes.split(","):
        c = cases[name]
        ref_seed = int(c["seed"][0])
        seeds = [ref_seed] + [s for s in range(1, args.seeds + 2) if s != ref_seed][:args.seeds - 1]
        for Np in [int(x) for x in args.np.split(",")]:
            for delay in args.delay.split(","):
                ests, stores = [], []
                for s in seeds:
                    est, st = run(c, s, tmpdir, ["-Np", str(Np), "-delay", delay] + args.extra.split())
                    if est is None:
                        print(name, Np, delay, s, "FAILED", st, flush=True)
                        continue
                    ests.append(est); stores.append(st)
                    print(name, Np, delay, "seed", s, "rho %.4e" % est[("Recomb", -1, -1, -1)][0], st, flush=True)
                for t in c["targets"]:
                    k = rb.target_key(t)
                    vals = [e.get(k, (float("nan"),))[0] for e in ests]
                    rows.append(dict(cls=name, Np=Np, delay=float(delay), target=rb.target_label(t), band=[t["min"], t["max"]], values=vals, stores=stores))
                    if t["type"] == "Recomb" or (t["type"] == "Coal" and t["epoch"] <= 1):
                        md.append("| %s | %d | %s | %s | %.4g – %.4g | %s | %s |" % (name, Np, delay, rb.target_label(t), t["min"], t["max"],
                                  " | ".join(("%.4g%s" % (v, "" if t["min"] <= v <= t["max"] else " **out**")) for v in vals),
                                  ", ".join("%d / %d" % (s["peak"], s["forced"]) for s in stores)))
                os.makedirs(os.path.dirname(args.out), exist_ok=True)
                json.dump(rows, open(args.out + ".json", "w"), indent=1)
                open(args.out + ".md", "w").write("\n".join(md) + "\n")


if __name__ == "__main__":
    main()
