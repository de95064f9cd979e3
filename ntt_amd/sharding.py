"""Batch sharding across ranks (one process per GPU).  Independent products shard with no
data-path collective: rank r owns a contiguous slice of the batch, exactly like the reference's
MPI variant gives rank r the primes i = r, r+size, ... (main_mpi_CRT.cc:470) and a contiguous
coefficient slice for the merge (main_mpi_CRT.cc:507-509).  The only communication is the
timing barrier / max-reduce of the benchmark and an optional gather of results."""


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` items owned by `rank` (sizes differ by at most one)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def crt_prime_owner(prime_index, world):
    """Rank that computes CRT prime `prime_index` in the reference's own placement (round-robin, main_mpi_CRT.cc:470,499).
    With four primes it keeps at most four ranks busy; crt_unit_range is the placement this package uses."""
    return prime_index % world


def crt_unit_range(batch, rank, world):
    """Contiguous run [lo, hi) of the 4 * batch residue products (unit = prime * batch + row) computed by `rank`.
    world = 1: everything; 2: two primes each; 4: one prime each; 8: one prime per pair of ranks, the batch halved inside
    the pair; any batch >= world / 4 keeps every rank busy."""
    return shard_range(4 * batch, rank, world)


def max_over_ranks(value, dist=None, device=None):
    """Max of a python float over all ranks (no-op without an initialised process group)."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local_rows, total, dist=None):
    """All-gather row shards (numpy arrays, possibly of unequal length) into the full batch on every rank."""
    import numpy as np
    if dist is None or not dist.is_initialized():
        return local_rows
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, local_rows)
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == total
    return out
