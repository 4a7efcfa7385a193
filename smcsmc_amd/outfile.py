"""The `.out` count table: writer mirroring PfParam::outFileHeader / appendToOutFile / FormatDouble
(/root/reference/src/pfparam.cpp:459-527) and CountModel::log_counts (count.cpp:66-158, prior
pseudo-counts count.cpp:161-227), plus a reader equivalent to the front-end's parse_outfile
(/root/reference/smcsmc/model.py:865-911) used to check the contract."""
import math
from collections import defaultdict


def format_double(d, scientific_bound=0.1, precision=2):
    """pfparam.cpp:482-497: fixed with `precision` decimals when scientific_bound < d < 10^(14-precision-1)
    or d == 0, otherwise scientific with 7 decimals; always right-aligned in 14 columns."""
    field_length = 14
    maxdouble = math.exp((field_length - precision - 1) * math.log(10.0))
    if d < maxdouble and (d > scientific_bound or d == 0.0):
        return "%*.*f" % (field_length, precision, d)
    return "%*.*e" % (field_length, field_length - 7, d)


HEADER = ("%6s %6s %14s %14s %6s %6s %6s %14s %14s %14s %14s %14s\n"
          % ("Iter", "Epoch", "Start", "End", "Type", "From", "To", "Opp", "Count", "Rate", "Ne", "ESS"))


def format_row(em_step, epoch, begin, end, event_type, from_pop, to_pop, opportunity, count, weight):
    ne = (opportunity + 1e-10) / (2.0 * count) if event_type == "Coal" else 0.0
    return ("%6d %6d %s %s %6s %6d %6d %s %s %s %s %s\n"
            % (em_step, epoch, format_double(begin), format_double(end), event_type, from_pop, to_pop,
               format_double(opportunity), format_double(count), format_double(count / (opportunity + 1e-10)),
               format_double(ne), format_double(1.0 / (weight / opportunity + 1e-10), 1.0, 3)))


def outfile_text(model, counts, np_particles, em_step=0):
    """The whole .out file from the counts of ParticleFilter.counts() (any number of populations)."""
    import numpy as np
    ct = list(model["change_times"])
    E = len(ct)
    P = int(model.get("n_pops", 1))
    ps = np.asarray(model["pop_sizes"], float).reshape(E, P)
    rho = model["recombination_rate"]
    end = lambda e: 1e+99 if e == E - 1 else ct[e + 1]      # noqa: E731
    co = np.asarray(counts["coal_opp"]).reshape(E, P)
    cc = np.asarray(counts["coal_count"]).reshape(E, P)
    cw = np.asarray(counts["coal_weight"]).reshape(E, P)
    out = [HEADER]
    for e in range(E):
        for a in range(P):
            out.append(format_row(em_step, e, ct[e], end(e), "Coal", a, -1, co[e, a] + 1.0,
                                  cc[e, a] + 1.0 / (2.0 * ps[e, a]), cw[e, a] + 1.0))
    ropp = rcount = rweight = 0.0
    for e in range(E):
        ropp += counts["rec_opp"][e] + 1.0
        rcount += counts["rec_count"][e] + rho
        rweight += counts["rec_weight"][e] + 1.0
    out.append(format_row(em_step, -1, 0.0, 1e+99, "Recomb", -1, -1, ropp, rcount, rweight))
    if P > 1:
        mr = np.asarray(model["mig_rates"], float).reshape(E, P, P)
        for e in range(E):
            for a in range(P):
                for b in range(P):
                    if a != b:
                        out.append(format_row(em_step, e, ct[e], end(e), "Migr", a, b, counts["mig_opp"][e, a] + 1.0,
                                              counts["mig_count"][e, a, b] + mr[e, a, b], counts["mig_weight"][e, a] + 1.0))
    dopp = counts["delayed_opp"]
    out.append(format_row(em_step, -1, 0.0, 1e+99, "Delay", -1, -1, dopp, counts["delayed_count"] / np_particles, dopp))
    out.append(format_row(em_step, -1, 0.0, 1e+99, "Resamp", -1, -1, dopp, counts["resample_count"], dopp))
    out.append(format_row(em_step, -1, 0, 1e+99, "LogL", -1, -1, 1.0, counts["logl"], 1.0))
    return "".join(out)


def _cxx_scientific(v, prec=5):
    """operator<< with std::scientific << setprecision(prec): at least two exponent digits"""
    return "%.*e" % (prec, v)


def recomb_text(local, nsam, iteration=0, start_position=1.0):
    """The `.recomb.gz` table of CountModel::dump_local_recomb_logs (count.cpp:616-654): one row per 100-bp interval,
    `iter locus size opp_per_nt <per-sample counts> time log_time`, the differential opportunity cumulated on the fly."""
    out = []
    if iteration == 0:
        out.append("iter\tlocus\tsize\topp_per_nt" + "".join("\t%d" % (k + 1) for k in range(nsam)) + "\ttime\tlog_time\n")
    cur = 0.0
    opp, cnt = local["opp_diff"], local["counts"]
    for idx in range(len(opp)):
        cur += opp[idx]
        row = ["%d" % iteration, "%.0f" % (idx * 100.0 + start_position), "%.0f" % 100.0, _cxx_scientific(cur / 100.0)]
        row += [_cxx_scientific(cnt[k][idx] / 100.0) for k in range(nsam + 2)]
        out.append("\t".join(row) + "\n")
    return "".join(out)


def descendants_text(mask):
    """print_descendants (descendants.hpp:51-65): '1' for each sample below the node, '0' for the ones before the last of
    them, nothing after it; '0' for the empty set."""
    mask = int(mask)
    if mask == 0:
        return "0"
    top = mask.bit_length()
    return "".join("1" if (mask >> i) & 1 else "0" for i in range(top))


def trees_text(kind, pos, height, desc, start_position=1.0, from_pop=None, to_pop=None):
    """The lines of `<prefix>.trees.gz` (ParticleContainer::printTrees, pc.cpp:515-555): event code (R, C, M), position
    (x + start_position - 1), height, from and to population, descendants; fixed notation with one decimal.  Without the
    population columns every coalescence is in population 0."""
    out = []
    for i, (k, x, t, d) in enumerate(zip(kind, pos, height, desc)):
        fr = -1 if k == 0 else (0 if from_pop is None else int(from_pop[i]))
        to = -1 if to_pop is None else int(to_pop[i])
        out.append("%s\t%.1f\t%.1f\t%d\t%d\t%s\n" % ("RCM"[int(k)], x + start_position - 1, t, fr, to, descendants_text(d)))
    return "".join(out)


def parse_outfile(path_or_text, is_text=False):
    """Reads a `.out` table back into summed statistics keyed ((Type, Epoch, From, To, -1), field) with fields Opp,
    Count, Wt, Start, End -- the dictionary the front-end's M-step works on (model.py:865-911).  Columns are found by
    their header names; a file holds one iteration; the weight column is rebuilt from the printed post-lag ESS as
    max(0, 1/ESS - 1e-10) * Opp (the inverse of how pfparam.cpp:524 prints it)."""
    text = path_or_text if is_text else open(path_or_text).read()
    rows = [ln.split() for ln in text.splitlines() if ln.strip()]
    column = {name: k for k, name in enumerate(rows[0])}
    body = rows[1:]
    if len({int(r[column["Iter"]]) for r in body}) > 1:
        raise ValueError("Found multiple iterations in .out file; expected only one")
    data = defaultdict(float)
    for r in body:
        stat = (r[column["Type"]], int(r[column["Epoch"]]), int(r[column["From"]]), int(r[column["To"]]), -1)
        opportunity = float(r[column["Opp"]])
        data[(stat, "Opp")] += opportunity
        data[(stat, "Count")] += float(r[column["Count"]])
        data[(stat, "Wt")] += max(0.0, 1.0 / float(r[column["ESS"]]) - 1e-10) * opportunity
        data[(stat, "Start")] = float(r[column["Start"]])
        data[(stat, "End")] = float(r[column["End"]])
        float(r[column["Rate"]]); float(r[column["Ne"]])          # present and numeric, not used
    return data
