"""In-process EM outer loop: chunk scheduler + reducer + M-step, the native counterpart of the reference's
per-iteration process farm (/root/reference/smcsmc/model.py:989-1184).

What it removes compared with the front-end: one process launch and `.seg` re-parse per chunk and iteration, the
1e6-tree lag calibration per chunk (it depends on the model only: done once per iteration), and the text round
trip of the sufficient statistics (the reduction is the ordered sum of smcsmc_amd.reduce, RCCL all-gather when
torch.distributed is initialised).

  PopulationModel          the fields of populationmodels.Population the path needs, front-end units
    .core_command_line()   argv of the next iteration's E-step (what populationmodels.py:300-437 builds)
    .device_model()        model tables for ParticleFilter (generations, per generation rates)
  counts_to_data()         .out rows as the front-end's parse_outfile reads them (model.py:865-911)
  m_step()                 Smcsmc.m_step, model.py:989-1048 (plain and variational-Bayes pseudo-counts)
  run_em()                 Smcsmc.do_iteration for all iterations, chunks sharded over ranks
"""
import copy

import numpy as np

from . import outfile, pf, recombguide, reduce as reducer


class PopulationModel:
    """Front-end units: times in 4*N0 generations, sizes relative to N0, migration rates 4*N0*m,
    mutation / recombination rates per bp per generation (populationmodels.py:26-70)."""

    def __init__(self, N0=10000, mutation_rate=2.5e-8, recombination_rate=1e-8, sequence_length=1e6, num_samples=2,
                 change_points=(0.0,), population_sizes=((1.0,),), num_populations=1, migration_rates=None,
                 sample_populations=None, migration_commands=None):
        self.N0 = N0
        self.mutation_rate = mutation_rate
        self.recombination_rate = recombination_rate
        self.sequence_length = sequence_length
        self.num_samples = num_samples
        self.num_populations = num_populations
        self.change_points = list(change_points)
        E, P = len(self.change_points), num_populations
        self.population_sizes = [list(row) if isinstance(row, (list, tuple)) else [row] * P for row in population_sizes]
        self.migration_rates = ([[list(r) for r in m] for m in migration_rates] if migration_rates is not None
                                else [[[0] * P for _ in range(P)] for _ in range(E)])
        self.sample_populations = list(sample_populations) if sample_populations is not None else [1] * num_samples
        self.migration_commands = list(migration_commands) if migration_commands is not None else [None] * E
        # event counts carried along for the variational-Bayes command line (populationmodels.py:259-268)
        self.population_event_counts = [[1e10] * P for _ in range(E)]
        self.migration_event_counts = [[[1e10] * P for _ in range(P)] for _ in range(E)]
        assert len(self.population_sizes) == E and len(self.migration_rates) == E and self.change_points[0] == 0.0

    # ---- the scrm-style command line of an E-step.  The token sequence and the number formatting are contract (the
    # binary parses them; the reference builds the same string in populationmodels.py:272-437 and
    # tests/golden/cmdlines.json holds its output); it is assembled here from per-epoch directives.
    def _sample_tokens(self):
        if self.num_populations == 1:
            return []
        per_pop = np.bincount(np.asarray(self.sample_populations, int), minlength=self.num_populations + 1)[1:]
        return ["-I", self.num_populations] + per_pop.tolist()

    def _size_tokens(self, epoch, vb):
        """-eN t x when all populations share a size, else one -en t pop x per population; -vb appends the event count."""
        when = self.change_points[epoch]
        sizes = self.population_sizes[epoch]
        counts = self.population_event_counts[epoch]
        if all(x == sizes[0] for x in sizes):
            return ["-eN", when, sizes[0]] + ([counts[0]] if vb else [])
        out = []
        for k, x in enumerate(sizes):
            out += ["-en", when, k + 1, x] + ([counts[k]] if vb else [])
        return out

    def _migration_tokens(self, epoch, vb):
        """-eM t M (M = (P-1) m) when every off-diagonal rate is m, else the full matrix with -ema."""
        P = self.num_populations
        if P == 1:
            return []
        when = self.change_points[epoch]
        rates = self.migration_rates[epoch]
        counts = self.migration_event_counts[epoch]
        off_diagonal = [rates[a][b] for a in range(P) for b in range(P) if a != b]
        if all(m == off_diagonal[0] for m in off_diagonal):
            # the reference's total includes the diagonal counts
            return ["-eM", when, off_diagonal[0] * (P - 1)] + ([sum(sum(row) for row in counts)] if vb else [])
        out = ["-ema", when]
        for a in range(P):
            for b in range(P):
                out.append(rates[a][b])
                if vb:
                    out.append(counts[a][b])
        return out

    def core_command_line(self, vb=False):
        scale = 4 * self.N0
        tokens = ["-N0", self.N0, "-t", scale * self.mutation_rate * self.sequence_length,
                  "-r", scale * self.recombination_rate * self.sequence_length, self.sequence_length]
        # the reference's template leaves the sample field empty for one population (two spaces in a row)
        head = " ".join(str(t) for t in tokens) + " " + " ".join(str(t) for t in self._sample_tokens()) + " "
        body = ["-vb"] if vb else []
        for epoch in range(len(self.change_points)):
            body += self._size_tokens(epoch, vb) + self._migration_tokens(epoch, vb)
            if self.migration_commands[epoch] is not None:
                body.append(self.migration_commands[epoch])
        return head + " ".join(str(t) for t in body)

    def device_model(self, lags=None, vb=False, **extra):
        """Model tables for ParticleFilter: generations and per-generation rates, -ej commands as joins."""
        E, P = len(self.change_points), self.num_populations
        ct = np.array(self.change_points, float) * 4 * self.N0
        ps = np.array(self.population_sizes, float).reshape(E, P) * self.N0
        m = dict(change_times=ct, pop_sizes=ps if P > 1 else ps[:, 0], nsam=self.num_samples,
                 loci_length=float(self.sequence_length), mutation_rate=self.mutation_rate,
                 recombination_rate=self.recombination_rate,
                 lags=np.zeros(E) if lags is None else np.asarray(lags, float))
        if P > 1:
            mr = np.array(self.migration_rates, float).reshape(E, P, P) / (4.0 * self.N0)
            sm = np.zeros((E, P, P))
            for i, cmd in enumerate(self.migration_commands):
                if not cmd:
                    continue
                tok = cmd.split()
                for k in range(0, len(tok), 4):
                    if tok[k] != "-ej":
                        raise ValueError("unsupported migration command: " + cmd)
                    a, b = int(tok[k + 2]) - 1, int(tok[k + 3]) - 1
                    sm[i, a, b] = 1.0
                    # scrm -ej also closes the joined population to migrants at that time; later epochs carry their
                    # own explicit matrices (same as the binary's flag parser, csrc/host/pfparam.cpp)
                    mr[i, :, a] = 0.0
                    mr[i, a, :] = 0.0
            m.update(n_pops=P, mig_rates=mr, single_mig=sm, sample_pops=[q - 1 for q in self.sample_populations])
        if vb:        # the event counts the -vb command line carries (populationmodels.py:335-398)
            m.update(vb_coal_counts=np.array(self.population_event_counts, float).reshape(E, P),
                     vb_mig_counts=np.array(self.migration_event_counts, float).reshape(E, P, P))
        m.update(extra)
        return m


def counts_to_data(model, counts, np_particles, rounded=True):
    """The sufficient statistics of one chunk keyed like Smcsmc.parse_outfile (model.py:865-911).  With
    rounded=True the values pass through the `.out` text form, so they equal what the front-end would read."""
    text = outfile.outfile_text(model, counts, np_particles)
    if rounded:
        return outfile.parse_outfile(text, is_text=True)
    # exact values: same keys, no text rounding
    E = len(model["change_times"])
    P = int(model.get("n_pops", 1))
    ps = np.asarray(model["pop_sizes"], float).reshape(E, P)
    data = {}
    for e in range(E):
        for a in range(P):
            key = ("Coal", e, a, -1, -1)
            data[(key, "Opp")] = float(np.asarray(counts["coal_opp"]).reshape(E, P)[e, a] + 1.0)
            data[(key, "Count")] = float(np.asarray(counts["coal_count"]).reshape(E, P)[e, a] + 1.0 / (2.0 * ps[e, a]))
    data[(("Recomb", -1, -1, -1, -1), "Opp")] = float(np.sum(counts["rec_opp"]) + E)
    data[(("Recomb", -1, -1, -1, -1), "Count")] = float(np.sum(counts["rec_count"]) + E * model["recombination_rate"])
    if P > 1:
        mr = np.asarray(model["mig_rates"], float).reshape(E, P, P)
        for e in range(E):
            for a in range(P):
                for b in range(P):
                    if a != b:
                        key = ("Migr", e, a, b, -1)
                        data[(key, "Opp")] = float(counts["mig_opp"][e, a] + 1.0)
                        data[(key, "Count")] = float(counts["mig_count"][e, a, b] + mr[e, a, b])
    data[(("LogL", -1, -1, -1, -1), "Count")] = float(counts["logl"])
    data[(("LogL", -1, -1, -1, -1), "Opp")] = 1.0
    return data


def add_data(total, data):
    """parse_outfile(..., data) accumulation over chunks (model.py:897-911)."""
    if total is None:
        return dict(data)
    for k, v in data.items():
        if k[1] in ("Start", "End"):
            total[k] = v
        else:
            total[k] = total.get(k, 0.0) + v
    return total


def _posterior_mean_rate(count, opportunity, prior):
    """Rate estimate of one event class from summed statistics: count / opportunity, with the Gamma(shape, rate) prior
    pseudo-counts of the variational-Bayes update folded in (model.py:997-1001: the shape adds to both).  Returns the
    rate and the count it was computed from."""
    rate_term, shape_term = prior
    count = count + shape_term
    return count / (opportunity + rate_term + shape_term), count


def m_step(pop, data, vb=False, vb_dirichlet=None, maxNE=1e99, infer_recomb=True):
    """The M-step of the front-end (Smcsmc.m_step, model.py:989-1048) on statistics summed over chunks: updates `pop`
    in place.  Ne = 1 / (2 * coalescence rate), capped at maxNE; migration rates go back to units of 4 N0 m; the
    counts behind every estimate are kept for the next -vb command line."""
    flat = (1e-30, 0.0)                          # plain EM: a vanishing guard against empty opportunities
    priors = {"Coal": flat, "Migr": flat}
    if vb:
        given = vb_dirichlet or {"ne": [1.0, 1.0], "migr": [1.0, 1.0]}
        priors = {"Coal": tuple(given["ne"]), "Migr": tuple(given["migr"])}
    P = pop.num_populations
    size_cap = maxNE / pop.N0
    for epoch in range(len(pop.change_points)):
        for a in range(P):
            stat = ("Coal", epoch, a, -1, -1)
            rate, count = _posterior_mean_rate(data[(stat, "Count")], data[(stat, "Opp")], priors["Coal"])
            pop.population_sizes[epoch][a] = min(size_cap, 1.0 / (2.0 * rate * pop.N0))
            pop.population_event_counts[epoch][a] = count
            for b in range(P):
                if b == a:
                    continue
                stat = ("Migr", epoch, a, b, -1)
                rate, count = _posterior_mean_rate(data[(stat, "Count")], data[(stat, "Opp")], priors["Migr"])
                pop.migration_rates[epoch][a][b] = rate * 4 * pop.N0
                pop.migration_event_counts[epoch][a][b] = count
    if infer_recomb:
        stat = ("Recomb", -1, -1, -1, -1)
        pop.recombination_rate = data[(stat, "Count")] / data[(stat, "Opp")]
    return pop


def run_em(pop, chunks, iterations, np_particles, seed=1, ess_fraction=0.5, lag_fraction=2.0, vb=False, maxNE=1e99,
           infer_recomb=True, rounded=True, device=0, rank=0, world=1, on_iteration=None, concurrent=4,
           alpha=0.0, beta=4.0, delay=0.5, guides=None):
    """EM over `chunks` (each a packable Segments object of smcsmc_amd.segments).  Chunks are sharded over ranks
    (reduce.assign_chunks, longest first); statistics are summed in chunk order on every rank, so all ranks take
    identical M-steps.  Up to `concurrent` chunks of a rank are filtered at the same time (one host thread and one
    stream pair each): a single chunk keeps only ~150 wavefronts busy, four chunks scale 4x on one MI355X.
    With `alpha` > 0 the local recombination map of every chunk becomes that chunk's recombination guide of the next
    iteration (Smcsmc.do_iteration, model.py:1129-1143: LocalRecombination(...).smooth(alpha, beta) -> `-guide`), kept
    in memory on the rank that owns the chunk; `guides` (dict chunk -> guide), if given, receives them.
    Returns (final model, list of per-iteration summed statistics)."""
    from concurrent.futures import ThreadPoolExecutor
    pop = copy.deepcopy(pop) if on_iteration is None else pop
    history = []
    sizes = [len(c) for c in chunks]
    mine = reducer.assign_chunks(sizes, world)[rank]
    guides = {} if guides is None else guides
    for it in range(iterations + 1):
        base = pop.device_model()
        lags = pf.calibrated_lags(base, lag_fraction=lag_fraction, device=device)     # model-only: once per iteration
        model = pop.device_model(lags=lags, vb=vb)
        def e_step(c):
            segs = chunks[c].pack(lags)
            m_c = model
            if c in guides:          # application delays: Model::lags_to_application_delays (smcsmc.cpp:306-307)
                m_c = dict(model, guide=guides[c], application_delays=lags / lag_fraction * delay)
            f = pf.ParticleFilter(m_c, np_particles, ess_fraction=ess_fraction, seed=seed + 1000 * it + c,
                                  max_trace_events=0, device=device, local_recomb=alpha > 0)
            f.load_segments(segs)
            f.init_prior(float(segs["start"][0]))
            f.run()
            f.finish()
            data = counts_to_data(model, f.counts(), np_particles, rounded=rounded)
            if alpha > 0 and it < iterations:
                lr = recombguide.LocalRecombination.from_filter(f.local_recomb(), pop.num_samples)
                guides[c] = lr.smooth(alpha, beta).guide(rounded=rounded)
            f.close()
            return c, data

        if concurrent > 1 and len(mine) > 1:
            with ThreadPoolExecutor(max_workers=min(concurrent, len(mine))) as pool:
                per_chunk = dict(pool.map(e_step, mine))
        else:
            per_chunk = dict(e_step(c) for c in mine)
        template = pf.unpack_counts(np.ones(pf.counts_len(len(pop.change_points), pop.num_populations)),
                                    len(pop.change_points), pop.num_populations)
        keys = sorted(counts_to_data(model, template, np_particles, rounded=rounded).keys())
        gathered = reducer.gather_chunk_data(per_chunk, len(chunks), keys=keys)    # every rank gets every chunk's statistics
        total = None
        for c in range(len(chunks)):
            total = add_data(total, gathered[c])
        history.append(total)
        if on_iteration is not None:
            on_iteration(it, pop, total)
        if it < iterations:
            m_step(pop, total, vb=vb, maxNE=maxNE, infer_recomb=infer_recomb)
    return pop, history
