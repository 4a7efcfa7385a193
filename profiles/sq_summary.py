#!/usr/bin/env python3
"""Sums the SQ counters of profiles/sq_counters.sh per kernel (rocprofv3 csv: one line per dispatch and counter)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    src, dst, what = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    files = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    dur = defaultdict(float)
    for f in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    out = {"workload": what, "kernels": {}}
    for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
        n = len(disp[k])
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
        out["kernels"][k] = dict(dispatches=n, us_per_dispatch=dur[k] / max(n, 1), per_dispatch={a: v / n for a, v in c.items()},
                                 share_of_wave_cycles=dict(parked_at_waitcnt=c.get("SQ_WAIT_ANY", 0) / wc, issue_stall=c.get("SQ_WAIT_INST_ANY", 0) / wc,
                                                           issuing=c.get("SQ_ACTIVE_INST_ANY", 0) / wc, issuing_valu=c.get("SQ_ACTIVE_INST_VALU", 0) / wc))
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
