#!/usr/bin/env python3
"""Where do the workgroup slots of a step go?  Runs C chunks of the bench workload through pf_run_many with the traced instance of the
row kernel (pf_set_wg_trace: a time stamp at either end of every workgroup) for a range of steps in the middle of the sweep and prints,
per role, when its workgroups start and how long they run, and how many workgroups are resident over the step.

    python profiles/wg_trace.py --chunks 8 [--debug BITS] [--first 20000 --steps 64] [--out file.json]

Roles by index within the chunk (run_sweep, pf_hip.hip): [0, nb) extend, nb bookkeeping, then the draw table (nb workgroups), the ledger
(nb + 192), the count columns from the oldest epoch to the youngest (KArgs::cw_off).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--length", type=float, default=2e7)
    ap.add_argument("--np", type=int, default=10000)
    ap.add_argument("--nsam", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=32)
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--count-wgs", type=int, default=-1)
    ap.add_argument("--count-workers", type=int, default=0)
    ap.add_argument("--first", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--out", default="")
    ap.add_argument("--raw", default="", help="save the raw trace (npz) for analysis elsewhere")
    a = ap.parse_args()
    a.debug |= 1 << 23          # PF_DEBUG_ONE_LAUNCH: the trace is of the form in which every role of a step is one launch (k_sweep4t / k_sweep4q)
    from smcsmc_amd import ParticleFilter
    wl = argparse.Namespace(np=a.np, nsam=a.nsam, length=a.length, epochs=a.epochs, seed=1, pops=1, uncalibrated_lags=False, host_data=False, device=0)
    cw = a.count_wgs if a.count_wgs >= 0 else (24 if a.chunks >= 6 else 0)
    chunks = []
    for k in range(a.chunks):
        model, segs = bench.build_workload(wl, seed=1 + k)
        f = ParticleFilter(model, a.np, ess_fraction=0.5, seed=1 + 1000 * k, max_trace_events=0, device=0, local_recomb=True, debug=a.debug, count_wgs=cw, count_workers=a.count_workers)
        f.load_segments(segs)
        chunks.append((f, segs))
    lead = chunks[0][0]

    def sweep():
        for f, sg in chunks:
            f.init_prior(float(sg["start"][0]))
        ParticleFilter.run_many([f for f, _ in chunks])
        for f, _ in chunks:
            f.finish()
    sweep()                                  # warm-up, untraced
    lead.set_wg_trace(a.first, a.steps)
    sweep()
    tr = lead.wg_trace()
    if a.raw:
        np.savez_compressed(a.raw, trace=tr, nb=(a.np + 255) // 256, chunks=a.chunks, count_wgs=cw, count_workers=a.count_workers, debug=a.debug)
    nb = (a.np + 255) // 256
    nL = nb + (32 if cw > 0 else 192)          # ledger workgroups per step (pf_hip.hip: 32 beside the particle blocks when the caller sets count_wgs)
    bounds = [("extend", 0, nb), ("bookkeeping", nb, nb + 1), ("draw table", nb + 1, 2 * nb + 1), ("ledger", 2 * nb + 1, 2 * nb + 1 + nL), ("counts", 2 * nb + 1 + nL, 1 << 30)]
    if a.count_workers > 0:
        bounds = bounds[:3] + [("workers", 2 * nb + 1, 1 << 30)]
    used = tr[:, :, 0] > 0
    t0 = np.where(used, tr[:, :, 0], np.uint64(1) << np.uint64(62)).min(axis=1)
    start = (tr[:, :, 0].astype(np.int64) - t0[:, None].astype(np.int64)) * 0.01          # us since the step's first workgroup started
    end = (tr[:, :, 1].astype(np.int64) - t0[:, None].astype(np.int64)) * 0.01
    bx = (tr[:, :, 3] & np.uint64(0xffffffff)).astype(np.int64)
    chunk = (tr[:, :, 3] >> np.uint64(32)).astype(np.int64)
    xcc = (tr[:, :, 2] >> np.uint64(32)).astype(np.int64) & 15
    span = np.where(used, end, 0).max(axis=1)
    res = {"chunks": a.chunks, "debug": a.debug, "count_wgs": cw, "count_workers": a.count_workers, "steps_traced": int(tr.shape[0]), "first_step": a.first,
           "step_span_us": {"mean": float(span.mean()), "min": float(span.min()), "max": float(span.max())},
           "workgroups_per_step": float(used.sum(axis=1).mean()), "roles": {}}
    print("steps %d  workgroups per step %.0f  span of a step (first start to last end) mean %.1f us  [%.1f, %.1f]" % (
        tr.shape[0], used.sum(axis=1).mean(), span.mean(), span.min(), span.max()))
    print("%-12s %8s %10s %10s %10s %10s %10s %12s" % ("role", "wgs/step", "start p50", "start p95", "dur p50", "dur p95", "end max", "wg-us/step"))
    for name, lo, hi in bounds:
        m = used & (bx >= lo) & (bx < hi)
        if not m.any():
            continue
        st, du, en = start[m], (end - start)[m], np.where(m, end, 0).max(axis=1)
        row = dict(wgs_per_step=float(m.sum(axis=1).mean()), start_p50=float(np.percentile(st, 50)), start_p95=float(np.percentile(st, 95)),
                   dur_p50=float(np.percentile(du, 50)), dur_p95=float(np.percentile(du, 95)), dur_max=float(du.max()), end_max_mean=float(en.mean()),
                   wg_us_per_step=float(du.sum() / tr.shape[0]))
        res["roles"][name] = row
        print("%-12s %8.0f %10.1f %10.1f %10.1f %10.1f %10.1f %12.0f" % (name, row["wgs_per_step"], row["start_p50"], row["start_p95"], row["dur_p50"], row["dur_p95"], row["end_max_mean"], row["wg_us_per_step"]))
    # extend role per chunk: when does each chunk's extend role start and end (its critical path)?
    print("extend role per chunk: mean start of its first workgroup / mean end of its last")
    per = []
    for c in range(a.chunks):
        m = used & (bx < nb) & (chunk == c)
        s0 = np.where(m, start, 1e9).min(axis=1).mean()
        e1 = np.where(m, end, 0).max(axis=1).mean()
        per.append((float(s0), float(e1)))
    print("  " + "  ".join("%d: %.1f-%.1f" % (c, p[0], p[1]) for c, p in enumerate(per)))
    res["extend_per_chunk"] = per
    # resident workgroups over the step, in bins of 4 us, by role
    edges = np.arange(0, max(8.0, span.max()) + 4.0, 4.0)
    occ = {}
    for name, lo, hi in bounds:
        m = used & (bx >= lo) & (bx < hi)
        o = []
        for b in edges[:-1]:
            mid = b + 2.0
            o.append(float((m & (start <= mid) & (end > mid)).sum() / tr.shape[0]))
        occ[name] = o
    res["resident_by_4us"] = {"t_us": edges[:-1].tolist(), **occ}
    print("resident workgroups (mean over the traced steps; 768 slots) at t =")
    print("  t(us)  " + " ".join("%5.0f" % (b + 2) for b in edges[:-1]))
    for name in occ:
        print("  %-6s " % name[:6] + " ".join("%5.0f" % v for v in occ[name]))
    print("  total  " + " ".join("%5.0f" % sum(occ[n][i] for n in occ) for i in range(len(edges) - 1)))
    # per XCD: workgroups, workgroup-microseconds and when its last workgroup ends (each XCD takes every eighth workgroup of the launch)
    hw = tr[:, :, 2] & np.uint64(0xffffffff)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    print("per XCD: workgroups per step, workgroup-us per step, mean end of its last workgroup, share of steps in which it ends last")
    last = np.stack([np.where(used & (xcc == x), end, 0).max(axis=1) for x in range(8)], axis=1)
    per_x = []
    for x in range(8):
        m = used & (xcc == x)
        per_x.append(dict(wgs=float(m.sum() / tr.shape[0]), wg_us=float(((end - start) * m).sum() / tr.shape[0]), last_end=float(last[:, x].mean()),
                          ends_last=float((last.argmax(axis=1) == x).mean())))
        print("  xcd %d: %6.0f %8.0f %7.1f %5.2f" % (x, per_x[-1]["wgs"], per_x[-1]["wg_us"], per_x[-1]["last_end"], per_x[-1]["ends_last"]))
    res["per_xcd"] = per_x
    print("  mean over steps of (earliest XCD end, latest XCD end): %.1f %.1f" % (last.min(axis=1).mean(), last.max(axis=1).mean()))
    # is the order of starts the order of the launch?  (rank correlation of start time with linear index, one step)
    lin = np.arange(tr.shape[1])
    k = tr.shape[0] // 2
    mu = used[k]
    order = np.argsort(start[k][mu], kind="stable")
    print("  step %d: workgroups that start before a workgroup with a lower linear index in the same XCD: %.3f" % (
        k, float(np.mean([(np.diff(lin[mu][np.argsort(start[k][mu] + 1e-9 * lin[mu])][xcc[k][mu][np.argsort(start[k][mu] + 1e-9 * lin[mu])] == x]) < 0).mean() for x in range(8)]))))
    ncu = len(set(zip(xcc[used].tolist(), se[used].tolist(), cu[used].tolist())))
    print("  distinct (xcd, se, cu) seen: %d" % ncu)
    res["xcc_share"] = [float((used & (xcc == x)).sum() / max(1, used.sum())) for x in range(8)]
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
