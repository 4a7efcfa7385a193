"""CountModel reduction across chunk ranks: the in-process replacement for the front-end's
file-based sum over chunks (/root/reference/smcsmc/model.py:1176-1184, 903-906).

One rank per GPU, each having filtered its own chromosome chunks.  The packed CountModel buffers
(a few KB) are all-gathered (RCCL over xGMI on the GPU box, gloo in the CPU tests) and summed in
rank order on every rank, so the result is bit-identical on all ranks and for any arrival order.
"""
import numpy as np

COUNT_KEYS = ("coal_count", "coal_opp", "coal_weight", "rec_count", "rec_opp", "rec_weight")
# structured models carry the migration statistics behind them, in the order of PF_COUNTS_LEN2 (include/smcsmc_pf.h):
#   coal_{count,opp,weight}[E][P]  rec_{count,opp,weight}[E]  mig_count[E][P][P]  mig_{opp,weight}[E][P]  4 scalars
MIGRATION_KEYS = ("mig_count", "mig_opp", "mig_weight")
SCALAR_KEYS = ("delayed_opp", "delayed_count", "resample_count", "logl")


def packed_length(E, P=1):
    """PF_COUNTS_LEN2(E, P)."""
    return 6 * E + 4 if P == 1 else 3 * E * P + 3 * E + E * P * P + 2 * E * P + 4


def pack_counts(counts):
    """The CountModel of one rank as the flat buffer of pf_get_counts (one or several populations: the migration
    statistics are part of it whenever the dictionary has them)."""
    keys = COUNT_KEYS + (MIGRATION_KEYS if "mig_count" in counts else ())
    return np.concatenate([np.asarray(counts[k], dtype=np.float64).reshape(-1) for k in keys] +
                          [[counts[k] for k in SCALAR_KEYS]])


def unpack_counts(packed, E, P=1):
    packed = np.asarray(packed, dtype=np.float64)
    assert len(packed) == packed_length(E, P), (len(packed), E, P)
    shapes = [("coal_count", (E, P)), ("coal_opp", (E, P)), ("coal_weight", (E, P)),
              ("rec_count", (E,)), ("rec_opp", (E,)), ("rec_weight", (E,))]
    if P > 1:
        shapes += [("mig_count", (E, P, P)), ("mig_opp", (E, P)), ("mig_weight", (E, P))]
    out, at = {}, 0
    for name, shape in shapes:
        size = int(np.prod(shape))
        block = packed[at:at + size].copy()
        out[name] = block if P == 1 else block.reshape(shape)
        at += size
    for name in SCALAR_KEYS:
        out[name] = float(packed[at]); at += 1
    return out


def allreduce_counts_ordered(packed, device=None):
    """Sum of the per-rank packed buffers in rank order (deterministic).  torch.distributed must be initialised."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.as_tensor(np.asarray(packed, dtype=np.float64), device=device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    total = gathered[0].clone()
    for r in range(1, world):
        total += gathered[r]
    return total.cpu().numpy()


def gather_chunk_data(per_chunk, n_chunks, keys=None, device=None):
    """Every rank contributes the statistics dictionaries of the chunks it filtered ({chunk id: {key: value}});
    every rank receives all of them.  One all-reduce of an [n_chunks, K] matrix in which each row has exactly one
    non-zero contributor (x + 0 is exact), so the result is bit-identical to a gather; without torch.distributed
    (single process) it is the identity.  `keys` fixes the column order (needed on ranks that own no chunk)."""
    try:
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except ImportError:
        active = False
    if not active:
        return [per_chunk[c] for c in range(n_chunks)]
    import torch
    if keys is None:
        keys = sorted(next(iter(per_chunk.values())).keys())
    mat = np.zeros((n_chunks, len(keys)))
    for c, d in per_chunk.items():
        mat[c] = [d[k] for k in keys]
    t = torch.as_tensor(mat, device=device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    mat = t.cpu().numpy()
    return [dict(zip(keys, mat[c].tolist())) for c in range(n_chunks)]


def assign_chunks(chunk_lengths, world):
    """Longest-first greedy assignment of chunks to ranks; returns rank -> [chunk ids]."""
    order = sorted(range(len(chunk_lengths)), key=lambda c: (-chunk_lengths[c], c))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for c in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(c)
        load[r] += chunk_lengths[c]
    return out
