#!/usr/bin/env python3
"""How the particle filter approaches the exact E-step of the two-sample model as the number of particles grows (GPU box).

    python tests/tools/hmm_convergence.py [--seeds 8] [--np 1000,4000,16000] [--lag 4] [--out gpurun_out/hmm_convergence]

Runs the reference's two-sample regression classes through bin/smcsmc (the drop-in binary: C++ host, C-ABI, HIP kernels) with
and without focused sampling, at a lag of `--lag` survival distances, and compares the mean estimates over seeds with
tests/golden/exact_hmm2.json (tests/exact_hmm2.py).  Writes <out>.json and <out>.md.
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import reference_bands as rb  # noqa: E402


def strip_focus(case):
    c = dict(case)
    toks, out, i = list(case["binary_argv"]), [], 0
    while i < len(toks):
        if toks[i] in ("-bias_heights", "-bias_strengths"):
            i += 1
            while i < len(toks) and not toks[i].startswith("-"):
                i += 1
            continue
        out.append(toks[i]); i += 1
    c["binary_argv"] = out
    return c


def run(case, focus, Np, lag, seeds, tmpdir, extra=()):
    c = case if focus else strip_focus(case)
    E = len([t for t in case["targets"] if t["type"] == "Coal"])
    vals = []
    for s in seeds:
        est = rb.run_case(c, s, tmpdir, ["-Np", str(Np), "-calibrate_lag", str(lag)] + list(extra))
        vals.append([est[("Coal", e, 0, -1)][0] for e in range(E)] + [est[("Recomb", -1, -1, -1)][0], est[("LogL", -1, -1, -1)][0]])
    return np.array(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--np", default="1000,4000,16000")
    ap.add_argument("--lag", type=float, default=4.0)
    ap.add_argument("--classes", default="TestConstPopSize,TestConstPopSize_FourEpochs")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hmm_convergence"))
    ap.add_argument("--extra", default="")
    args = ap.parse_args()
    gold = json.load(open(os.path.join(ROOT, "tests/golden/exact_hmm2.json")))["classes"]
    cases = {c["name"]: c for c in rb.load_cases(variants=False)}
    tmpdir = tempfile.mkdtemp(prefix="hmmconv_")
    seeds = list(range(1, args.seeds + 1))
    rows, md = [], ["| class | focusing | Np | " + "quantity | exact | mean over %d seeds | s.e. | relative difference |" % len(seeds), "|---|---|---|---|---|---|---|---|"]
    for name in args.classes.split(","):
        ex = gold[name]["exact"]
        exact = np.array(ex["ne"] + [ex["rho"], ex["logl"]])
        E = len(ex["ne"])
        labels = ["Ne epoch %d" % e for e in range(E)] + ["rho", "log-likelihood"]
        for focus in (False, True):
            for Np in [int(x) for x in args.np.split(",")]:
                v = run(cases[name], focus, Np, args.lag, seeds, tmpdir, args.extra.split())
                mean, se = v.mean(0), v.std(0, ddof=1) / np.sqrt(len(seeds))
                rel = mean / exact - 1.0
                rows.append(dict(cls=name, focus=focus, Np=Np, lag=args.lag, labels=labels, exact=exact.tolist(), mean=mean.tolist(),
                                 se=se.tolist(), rel=rel.tolist(), values=v.tolist()))
                for k, lab in enumerate(labels):
                    md.append("| %s | %s | %d | %s | %.6g | %.6g | %.2g | %+.2e |" % (name, "on" if focus else "off", Np, lab, exact[k], mean[k], se[k], rel[k]))
                print(name, "focus" if focus else "plain", Np, " ".join("%+.2f%%" % (100 * r) for r in rel[1:-1]), "logl %+.1e" % rel[-1], flush=True)
                os.makedirs(os.path.dirname(args.out), exist_ok=True)
                json.dump(rows, open(args.out + ".json", "w"), indent=1)
                open(args.out + ".md", "w").write("\n".join(md) + "\n")


if __name__ == "__main__":
    main()
