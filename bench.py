#!/usr/bin/env python3
"""bench.py -- genome segments / second of the particle-filter forward sweep on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  * a "step" = one E-step sweep of the particle filter over one synthetic chromosome chunk
    (the do-while of /root/reference/src/smcsmc.cpp:324-360 plus the final flush 371-373),
    inputs (segments, model tables) already resident in HBM when the timed region starts.
  * workload at N=1: BASELINE.json configs[2] -- 2 diploids (4 haplotypes), 100 Mb, Np=10000,
    E=32 epochs, N0=1e4, mu=2.5e-8, rho=1e-8, data from the in-repo SMC'
    simulator on the device (k_simulate, seeded; --host-data: the numpy one of round 1).  N>1: one such chunk per GPU (weak scaling: chunks are independent,
    smcsmc/model.py:563-662), then one RCCL all-gather of the packed CountModel buffers summed
    in rank order (deterministic), the functional equivalent of smcsmc/model.py:1176-1184.
  * value = (segments processed by all ranks over K steps) / (max over ranks of the timed wall time).
Extra objects: "roofline" for the dominant kernel (the row kernel of the extend role: k_sweep4 for the default workload, one launch per
row on the filter stream; the ledger and count roles run beside it as k_sweep_blc4 on the counting stream) and "cpu_baseline" (the CPU
oracle, kind "port": the reference binary cannot be built here) timed on a bounded sample on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
_LAG_CACHE = {}


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def workload_label(args):
    return "Np=%d, %s, %.0f Mb" % (args.np, "%d diploids" % (args.nsam // 2) + ("" if args.pops == 1 else " from %d populations" % args.pops),
                                  args.length / 1e6)


def build_workload(args, seed):
    from smcsmc_amd import segments as segmod
    from smcsmc_amd import simulate
    n, L, E = args.nsam, float(args.length), args.epochs
    N0, mu, rho = 1e4, 2.5e-8, 1e-8
    ct = simulate.default_epochs(E)
    ps = np.full(E, N0)
    # the reference's uncalibrated per-epoch lag 4/(rho*top_t) (count.cpp:240-245); replaced below by the calibrated
    # lags the binary uses by default (-calibrate_lag 2: twice the median survival distance, smcsmc.cpp:169-263,
    # count.cpp:261-265), whose calibration is timed separately
    lags = np.array([4.0 / (rho * (ct[e + 1] if e + 1 < E else ct[-1])) for e in range(E)]) if E > 1 else np.array([20000.0])
    model = dict(change_times=ct, pop_sizes=ps, lags=lags, nsam=n, loci_length=L, mutation_rate=mu,
                 recombination_rate=rho)
    if args.pops > 1:
        P = args.pops
        split = int(np.searchsorted(ct, 0.5 * 4 * N0))          # first epoch that starts at or after the split
        split = min(max(split, 1), E - 1)
        mr = np.zeros((E, P, P)); sm = np.zeros((E, P, P))
        for e in range(split):
            mr[e] = (1.0 / (4 * N0)) * (1 - np.eye(P))
        sm[split, 1:, 0] = 1.0
        model.update(n_pops=P, pop_sizes=np.repeat(ps[:, None], P, axis=1), mig_rates=mr, single_mig=sm,
                     sample_pops=[i * P // n for i in range(n)])
    if getattr(args, "host_data", False):
        # the numpy simulator of smcsmc_amd/simulate.py (independent of the device code; minutes for 100 Mb, cached)
        cache = "/tmp/smcsmc_bench_n%d_L%d_E%d_s%d.npz" % (n, int(L), E, seed)
        if os.path.exists(cache):
            z = np.load(cache)
            seg = {k: z[k] for k in ("start", "length", "alleles")}
        else:
            seg = simulate.simulate_seg(n, L, mu, rho, ct, ps, seed=seed)
            np.savez(cache, **seg)
    else:
        # generated on the device (k_simulate / k_simulate_mp: one lane per chunk, the filter's own SMC' transition, from
        # the model the sweep then assumes -- with several populations the isolation-with-migration model), seeded
        seg = simulate.simulate_seg_device(n, L, mu, rho, ct, ps, seed=seed, nchunks=1, device=getattr(args, "device", 0),
                                           structure=model if args.pops > 1 else None)[0]
    max_seg_len = int(2.0 / (rho * 4 * N0))        # pfparam.cpp:364
    S = segmod.Segments.from_sites(seg["start"], seg["length"], seg["alleles"], n, L, max_segment_length=max_seg_len)
    if not getattr(args, "uncalibrated_lags", False):
        from smcsmc_amd import pf as pfmod
        key = (n, E, args.pops)
        if key not in _LAG_CACHE:          # model-only: once per process
            _LAG_CACHE[key] = pfmod.calibrated_lags(model, lag_fraction=2.0, device=getattr(args, "device", 0))
        lags = _LAG_CACHE[key]
        model["lags"] = lags
    return model, S.pack(lags)


def run_sweep(pf, segs):
    pf.init_prior(float(segs["start"][0]))
    pf.run()
    pf.finish()


def cpu_baseline(args, model, segs):
    """The CPU oracle (restatement of the reference's single-threaded path) on a bounded prefix."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build()
    nseg = min(args.cpu_segments, len(segs["start"]))
    sub = {k: v[:nseg] for k, v in segs.items()}
    o = oracle_lib.Oracle(model, args.np, ess_fraction=0.5, seed=args.seed, max_trace_events=0)
    o.init_prior(float(sub["start"][0]))
    si = o.pack_segments(model, sub)
    t0 = time.perf_counter()
    done = 0
    spent_counting = 0.0
    distinct = []
    for s in range(nseg):
        o.update_segment(si, s)
        pos = min(sub["start"][s] + sub["length"][s], model["loci_length"])
        o.count(pos)
        o.resample(pos)
        done += 1
        if s % 25 == 24:
            # particles in distinct states: what the reference's multiplicity records (particle.cpp:831-857) would hold
            # instead of Np expanded copies (not part of the timed work)
            tc = time.perf_counter()
            pa = o.particles()
            distinct.append(len(np.unique(pa["heights"].reshape(args.np, -1), axis=0)))      # distinct local trees
            spent_counting += time.perf_counter() - tc
        if time.perf_counter() - t0 - spent_counting > args.cpu_seconds:
            break
    dt = time.perf_counter() - t0 - spent_counting
    st = o.stats()
    o.close()
    return {"value": done / dt, "unit": "segments/s", "cores": 1, "kind": "port", "cpu": cpu_model_name(),
            "sample": "first %d segments of the same workload (Np=%d), %.1f s, single thread, oracle/libsmc_oracle.so "
                      "(-O3 -DNDEBUG); %d genealogy updates; %.0f distinct local trees among the %d particles on average "
                      "(a lower bound on the records the reference's multiplicity representation keeps; the port expands them)"
                      % (done, args.np, dt, st["recombinations"], float(np.mean(distinct)) if distinct else float(args.np), args.np)}


def cpu_worker(path):
    """Child process of cpu_baseline_all_cores: the same bounded oracle run on a saved prefix; prints one JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pickle
    import oracle_lib
    d = pickle.load(open(path, "rb"))
    model, sub, Np, seed, seconds = d["model"], d["segs"], d["np"], d["seed"], d["seconds"]
    o = oracle_lib.Oracle(model, Np, ess_fraction=0.5, seed=seed, max_trace_events=0)
    o.init_prior(float(sub["start"][0]))
    si = o.pack_segments(model, sub)
    t0 = time.perf_counter()
    done = 0
    for s in range(len(sub["start"])):
        o.update_segment(si, s)
        pos = min(sub["start"][s] + sub["length"][s], model["loci_length"])
        o.count(pos)
        o.resample(pos)
        done += 1
        if time.perf_counter() - t0 > seconds:
            break
    print(json.dumps({"done": done, "dt": time.perf_counter() - t0}))


def cpu_baseline_all_cores(args, model, segs):
    """How the reference's front-end uses a host: one single-threaded process per chunk, all at once
    (model.py:1094-1098).  One oracle process per core, each on the same prefix with its own seed; the processes never
    touch the GPU (plain children started with subprocess, nothing is exec'ed from this process)."""
    import pickle
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build()
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))      # a one-GPU box of this pool has a 16-core share
    nseg = min(args.cpu_segments, len(segs["start"]))
    sub = {k: v[:nseg] for k, v in segs.items()}
    procs = []
    with tempfile.TemporaryDirectory() as tmp:
        for c in range(cores):
            path = os.path.join(tmp, "w%d.pkl" % c)
            pickle.dump({"model": model, "segs": sub, "np": args.np, "seed": args.seed + c, "seconds": args.cpu_seconds},
                        open(path, "wb"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        rates = []
        for pr in procs:
            out, _ = pr.communicate(timeout=args.cpu_seconds * 4 + 120)
            try:
                r = json.loads(out.strip().splitlines()[-1])
                rates.append(r["done"] / r["dt"])
            except Exception:
                pass
    return {"value": float(sum(rates)), "unit": "segments/s", "cores": len(rates), "kind": "port", "cpu": cpu_model_name(),
            "sample": "%d concurrent single-threaded oracle processes, each the first segments of the same workload "
                      "(Np=%d) for %.0f s; aggregate rate" % (len(rates), args.np, args.cpu_seconds)}


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--np", "--particles", dest="np", type=int, default=10000,
                    help="particles per chunk (--particles: the spelling to use behind torch.distributed.run, whose own parser trips over --np)")
    ap.add_argument("--nsam", type=int, default=4)
    ap.add_argument("--length", type=float, default=1e8)
    ap.add_argument("--epochs", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pops", type=int, default=1,
                    help="populations: >1 switches to the isolation-with-migration shape of BASELINE.json configs[4] "
                         "(samples split evenly, symmetric migration 4*N0*m = 1, all populations join at 0.5*4N0)")
    ap.add_argument("--cpu-segments", type=int, default=2000)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the one-oracle-process-per-core baseline")
    ap.add_argument("--timing-period", type=int, default=16, help="time every k-th segment's kernels with HIP events")
    ap.add_argument("--no-local-recomb", action="store_true",
                    help="do not record the 100-bp local recombination map (the binary always records it, smcsmc.cpp:376-383)")
    ap.add_argument("--host-data", action="store_true",
                    help="synthetic data from the numpy simulator (round-1 inputs) instead of the device-side simulator")
    ap.add_argument("--uncalibrated-lags", action="store_true",
                    help="the reference's uncalibrated lags 4/(rho*top_t) instead of the calibrated default of the binary")
    ap.add_argument("--debug", type=int, default=0, help="pf_params.debug bits (include/smcsmc_pf.h): 4 no counting, 8 two launches per row, 16 the round-2 row paths, 32 no speculative staging, 64 extend role and the other roles as two launches (round 3 form), 8388608 every role of a step in one launch (rounds 3-4)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single-GPU box: every rank uses device 0 and the collectives go through gloo "
                         "on host tensors (RCCL refuses two ranks on one device); the numbers it prints are not a scaling result")
    ap.add_argument("--chunks-per-gpu", type=int, default=1,
                    help="independent chunks filtered concurrently on each GPU, all in one launch per row (pf_run_many); "
                         "1 = the headline single-chunk configuration")
    ap.add_argument("--count-wgs", type=int, default=0, help="pf_params.count_wgs: count workgroups per epoch in the row pipeline (0 = default: one per "
                    "particle block; 24 with six or more chunks per GPU, where the chip holds a fraction of the count workgroups at a time)")
    ap.add_argument("--count-workers", type=int, default=-1, help="pf_params.count_workers: with several chunks per GPU, workgroups per chunk and step that take the "
                    "ledger and count work off a queue (0 = every work item a workgroup of the launch; default: 0 for one chunk)")
    ap.add_argument("--log-cap", type=int, default=-1, help="pf_params.log_cap: event-log records per particle slot (0 = the library's default, 16384; default here: 4096 "
                    "for one population -- the C3 workload needs between 1024 and 2048, too small a ring is a reported error -- because the 10 GB of the default ring "
                    "cost 2.4 %% in address translation: 3.27e4 -> 3.35e4 segments/s)")
    ap.add_argument("--chunk-threads", action="store_true",
                    help="with --chunks-per-gpu: one host thread and stream per chunk (rounds 1 and 2) instead of pf_run_many")
    args = ap.parse_args()

    if args.log_cap < 0:
        args.log_cap = 4096 if args.pops == 1 else 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    cdev = "cuda"                       # where the tensors of the collectives live
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
            cdev = "cpu"
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from smcsmc_amd import ParticleFilter
    from smcsmc_amd import build as pfbuild
    from smcsmc_amd import pf as pfmod
    if not os.path.exists(pfmod.LIB_PATH):
        pfbuild.build_lib()

    import threading
    C = max(1, args.chunks_per_gpu)
    # six or more chunks per GPU: count_wgs set (to the most a column can have) makes the library taper the columns of the young epochs and trim its
    # ledger workgroups -- 8 chunks 1.09e5 -> 1.17e5 segments/s, 12 chunks 1.11e5 -> 1.29e5; fewer chunks are faster without (4: 8.95e4 against 8.55e4)
    many_wgs = (args.np + 255) // 256
    dev = local_rank if (world > 1 and not args.rehearse_on_one_gpu) else 0
    args.device = dev
    chunks = []
    for k in range(C):
        model, segs = build_workload(args, seed=args.seed + rank * C + k)     # independent chunks
        f = ParticleFilter(model, args.np, ess_fraction=0.5, seed=args.seed + 1000 * (rank * C + k), max_trace_events=0,
                           device=dev, local_recomb=not args.no_local_recomb, debug=args.debug,
                           count_wgs=max(0, args.count_wgs) or (many_wgs if (args.chunks_per_gpu >= 6 and args.count_wgs == 0) else 0),
                           count_workers=max(0, args.count_workers), log_cap=args.log_cap)
        f.load_segments(segs)
        chunks.append((f, segs))
    pf, segs = chunks[0]
    n_segments = sum(len(sg["start"]) for _, sg in chunks)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    many = C > 1 and not args.chunk_threads and args.pops == 1 and args.nsam <= 8 and not (args.debug & (8 | 16))

    def sweep_all():
        if C == 1:
            run_sweep(pf, segs)
            return
        if many:
            for f, sg in chunks:
                f.init_prior(float(sg["start"][0]))
            ParticleFilter.run_many([f for f, _ in chunks])
            for f, _ in chunks:
                f.finish()
            return
        th = [threading.Thread(target=run_sweep, args=(f, sg)) for f, sg in chunks]
        for t in th:
            t.start()
        for t in th:
            t.join()

    for _ in range(args.warmup):
        sweep_all()
    for f, _ in chunks:
        f.sync()
    pf.set_timing(args.timing_period)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep_all()
    for f, _ in chunks:
        f.sync()
    barrier()
    dt = time.perf_counter() - t0
    kt = pf.kernel_times()
    st = pf.stats()
    logl = sum(f.logl() for f, _ in chunks)
    counts = pf.counts()
    for f, _ in chunks[1:]:
        cc = f.counts()
        for kk in counts:
            counts[kk] = counts[kk] + cc[kk]
    n_seg0 = len(segs["start"])

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_max = float(tmax.item())
        segs_all = torch.tensor([float(n_segments)], dtype=torch.float64, device=cdev)
        dist.all_reduce(segs_all, op=dist.ReduceOp.SUM)
        total_segments = float(segs_all.item())
        # CountModel "all-reduce": all-gather + sum in rank order (bit-identical for any arrival order), in the packed layout
        # of the library (reduce.pack_counts = PF_COUNTS_LEN2: what bin/smcsmc exchanges and the gloo tests cover)
        from smcsmc_amd import reduce as pfreduce
        counts["logl"] = logl
        packed = pfreduce.pack_counts(counts)
        mine = torch.tensor(packed, dtype=torch.float64, device=cdev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        reduced = gathered[0].clone()
        for r in range(1, world):
            reduced += gathered[r]
        logl_sum = float(reduced[-1].item())
    else:
        dt_max = dt
        total_segments = float(n_segments)
        logl_sum = logl

    if rank == 0:
        value = total_segments * args.steps / dt_max
        ext_ms, ext_launches = kt["extend"]
        avg_ext_us = 1e3 * ext_ms / max(1, ext_launches)
        rec_per_seg = st["records"] / max(1, n_seg0)                # records appended per segment (last sweep, chunk 0)
        state_bytes = st["state_bytes_per_particle"]
        rec_bytes = 8 * (5 + args.nsam - 1)
        # algorithmic bytes of one k_extend launch: every particle's state read and written once,
        # plus the event records appended
        alg_bytes = args.np * 2 * state_bytes + rec_per_seg * rec_bytes
        # Duration of one launch of the row kernel.  HIP events bracket every `--timing-period`-th launch (avg_ext_us: an estimate
        # from a sample, minus the cost of an empty span); the driver's clock sees rows, not launches, so the figure the roofline
        # fraction is computed from is the wall-clock time per row of chunk 0's sweep -- ms_per_step / rows, launch gaps included,
        # an upper bound of the kernel's duration that agrees with rocprofv3's per-kernel average (profiles/round4) -- whenever one
        # launch per row is what runs (one chunk per GPU); with several chunks in flight the sampled event figure stands.
        us_per_row_wall = 1e6 * (dt_max / args.steps) / max(1, n_seg0)
        one_launch_per_row = C == 1 and args.nsam <= 8 and not (args.debug & (8 | 3))
        dur_us = us_per_row_wall if (one_launch_per_row and args.pops == 1) else avg_ext_us
        achieved = alg_bytes / (dur_us * 1e-6) / 1e9 if dur_us > 0 else 0.0
        # HBM traffic of the same kernel from the PMC counters: collected by profiles/collect_round*.sh in separate
        # rocprofv3 passes (counters cannot be read from inside this process) and kept under profiles/; a file counts only when
        # it is of this shape AND of the kernel this run's rows go through
        if args.pops == 1:
            kern_key = "k_pipe" if args.debug & 16 else ("k_row" if args.debug & 8 else "k_sweep")
        else:
            kern_key = "k_sweep_xmp" if (args.nsam <= 8 and not (args.debug & (16 | 3))) else ("k_extend_mpr" if args.nsam <= 8 and not (args.debug & 3) else "k_extend_mp")
        traffic, traffic_src = None, None
        import glob
        for rnd in ("round2", "round3", "round4"):                       # the newest matching file wins
            for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "*_pmc_k_*.json"))):
                pmc = json.load(open(pmc_path))
                if pmc.get("shape") == {"nsam": args.nsam, "np": args.np, "epochs": args.epochs, "pops": args.pops} and kern_key + "<" in pmc.get("kernel", "").replace("4<", "<"):
                    traffic = pmc["traffic_bytes_per_launch"]
                    traffic_src = dict(file=os.path.relpath(pmc_path, ROOT), kernel=pmc.get("kernel"), counter_run=pmc.get("counter_run", pmc.get("note", ""))[:300])
        out = {
            "metric": "genome segments/sec per EM iteration (%s)" % workload_label(args),
            "value": value, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d haplotypes, %s%.0f Mb, Np=%d, E=%d epochs, one chunk per GPU"
                                   % (args.nsam, "" if args.pops == 1 else "%d-population isolation-with-migration model, " % args.pops,
                                      args.length / 1e6, args.np, args.epochs),
                       "populations": args.pops, "local_recombination_map": not args.no_local_recomb,
                       "segments_per_chunk": n_segments, "nsam": args.nsam, "np": args.np,
                       "sequence_length": args.length, "epochs": args.epochs, "event_log_records_per_particle": args.log_cap or 16384,
                       "count_workgroups_per_epoch": ("one per particle block, tapered for the young epochs" if (args.chunks_per_gpu >= 6 and args.count_wgs == 0) else (args.count_wgs if args.count_wgs > 0 else "one per particle block")),
                       "parallelism": "%d chunk(s) per gpu%s x %d gpu(s)" % (C, ((", two launches per row for all of them" if (args.nsam <= 4 and not (args.debug & (1 << 23))) else ", one launch per row for all of them") if many else ", one host thread and stream each") if C > 1 else "", world), "log_likelihood_sum": logl_sum},
            "roofline": {"bound": "hbm", "kernel": ("k_pipe (one launch per row: extend workgroups; bookkeeping, ledger and counts ride along)" if args.debug & 16 else ("k_sweep4 (one launch per row: the extend, bookkeeping and draw roles; ledger and counts as k_sweep_blc4 on a second stream)" if args.nsam <= 4 and not (args.debug & (1 << 23)) else "k_sweep (one launch per row: extend workgroups; bookkeeping, ledger and counts ride along)")) if args.pops == 1 and args.nsam <= 8 and not (args.debug & 8) else ("k_row" if args.pops == 1 and args.nsam <= 8 else (("k_sweep_xmp (extend role of the row pipeline; bookkeeping, ledger and counts as k_sweep_blc on a second stream)" if not (args.debug & 16) else "k_extend_mpr (register tree, completes the previous row while loading)") if args.nsam <= 8 and not (args.debug & 3) else "k_extend_mp") if args.pops > 1 else "k_extend"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": alg_bytes,
                         "alg_bytes_model": "2 x %d B of state per particle (read and written once per row) x Np + %.2f records of %d B appended per row (DESIGN.md section 2)" % (state_bytes, rec_per_seg, rec_bytes),
                         "launch_us": dur_us,
                         "launch_us_basis": ("wall clock per row (ms_per_step / rows of the chunk): every launch, gaps included" if dur_us == us_per_row_wall
                                             else "HIP events on every %d-th launch" % args.timing_period),
                         "avg_launch_us": avg_ext_us, "us_per_row_wall": us_per_row_wall,
                         # SURVEY.md section 8(d) prices a row at 11 MB with its model of the state (256 B per particle) and of the records
                         # (32-byte payloads, one per tree slice and epoch): against that figure the fraction would be
                         "frac_against_survey_model_bytes": (11.0e6 / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (args.nsam == 4 and args.np == 10000 and args.pops == 1 and dur_us > 0) else None,
                         "kernel_ms_estimate": {k: v[0] for k, v in kt.items()},
                         "kernel_launches": {k: v[1] for k, v in kt.items()}},
        }
        try:
            # achievable HBM bandwidth on this device, for scale: a 1 GiB device-to-device copy (bytes read + written)
            src = torch.empty(1 << 27, dtype=torch.float64, device="cuda:%d" % dev).fill_(1.0)
            dst = torch.empty_like(src)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            best = 1e30
            for _ in range(4):
                e0.record(); dst.copy_(src); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            out["roofline"]["measured_copy_GBs"] = 2 * src.numel() * 8 / (best * 1e-3) / 1e9
            del src, dst
        except Exception:
            pass
        if world == 1:
            # the lag calibration (calculate_median_survival_distances, smcsmc.cpp:169-263: prior ARGs until every
            # epoch has 200 survival distances) is model-only work outside the sweep: timed separately, and an
            # all-in rate with it included, as the reference pays it once per chunk and E-step
            from smcsmc_amd import pf as _pf
            tc0 = time.perf_counter()
            _, cal_trees = _pf.median_survival(model, seed=1, device=dev)
            cal_s = time.perf_counter() - tc0
            sweep_s = dt_max / args.steps
            out["config"]["lag_calibration_ms"] = 1e3 * cal_s
            out["config"]["lag_calibration_trees"] = int(cal_trees)
            out["config"]["all_in_segments_per_s"] = total_segments / (sweep_s + cal_s)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, model, segs)
            if not args.no_cpu_all_cores:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, model, segs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
