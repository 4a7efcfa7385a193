#!/usr/bin/env python3
"""Reduces the raw rocprofv3 output of profiles/collect_round2.sh to the small per-kernel summaries kept under
profiles/ (the raw traces are tens of MB per pass): usage summarize.py <gpurun_out/prof_TAG prefix> <out prefix>.

  <out>_kernel_stats.csv      rocprofv3 --kernel-trace --stats: calls, total and average duration per kernel
  <out>_pmc_FETCH_SIZE.csv    mean / sum of the counter per kernel (one --pmc pass each: the two do not fit one pass)
  <out>_pmc_WRITE_SIZE.csv
  <out>_pmc.json              HBM bytes per launch of the dominant kernel from the two counters
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def pmc_summary(path, out):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"], row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    with open(out, "w") as f:
        f.write("kernel_name,counter,dispatches,avg_value,sum_value\n")
        for (name, counter), (n, total) in rows:
            f.write('"%s",%s,%d,%r,%r\n' % (name, counter, n, total / n, total))
    return {name: (n, total / n) for (name, counter), (n, total) in rows}


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    stats = glob.glob(src + "_stats/*kernel_stats.csv")
    if stats:
        shutil.copyfile(stats[0], out + "_kernel_stats.csv")
    res = {}
    for counter, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        files = glob.glob(src + "_%s/*counter_collection.csv" % tag)
        if files:
            res[counter] = pmc_summary(files[0], out + "_pmc_%s.csv" % counter)
    if len(res) == 2:
        # one file per row kernel (everything whose name starts with k_sweep / k_pipe / k_extend / k_row), named after it; `shape` and
        # the note of the counter run (argv[3], argv[4]) let bench.py pick the file of its own workload AND kernel
        shape = json.loads(sys.argv[3]) if len(sys.argv) > 3 else None
        run_note = sys.argv[4] if len(sys.argv) > 4 else ""
        for name in res["WRITE_SIZE"]:
            short = name.replace("void ", "").split("<")[0].split("(")[0]
            if not short.startswith(("k_sweep", "k_pipe", "k_extend", "k_row")) or short == "k_sweep_seed":
                continue
            fr, wr = res["FETCH_SIZE"].get(name), res["WRITE_SIZE"].get(name)
            if not (fr and wr):
                continue
            doc = {
                "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes), profiles/collect_round4.sh",
                "kernel": name, "dispatches": fr[0], "fetch_size_kb_avg_raw": fr[1], "write_size_kb_avg": wr[1], "fetch_correction": 2.0,
                "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads "
                        "(MI355X_MICROARCH.md, HBM); the loads here are 8 B per lane or scattered 64-B records, for which the "
                        "counter is uncalibrated: the corrected figure (x2) is an upper bound, the raw one a lower bound. "
                        "WRITE_SIZE is exact." + ("  The launch also carries the bookkeeping, ledger and count workgroups."
                                                  if short in ("k_pipe", "k_sweep", "k_sweep4") else ""),
                "counter_run": run_note,
                "traffic_bytes_per_launch": 1024.0 * (2.0 * fr[1] + wr[1]),
                "traffic_bytes_per_launch_lower": 1024.0 * (fr[1] + wr[1]),
            }
            if shape:
                doc["shape"] = shape
            json.dump(doc, open(out + "_pmc_%s.json" % short, "w"), indent=1)
    # the raw traces stay on the box
    for d in glob.glob(src + "_stats") + glob.glob(src + "_fetch") + glob.glob(src + "_write"):
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
