"""Large-modulus products with the CRT primes spread over ranks (one process per GPU).

Mirrors the reference's MPI variant (main_mpi_CRT.cc:464-529): rank r runs the residue
products of primes i = r, r+world, ...; the residue vectors are exchanged (there: one
`MPI_Bcast` per prime, here one broadcast per prime over RCCL / gloo); every rank merges
a contiguous coefficient slice; the slices are combined with a sum of zero-padded
buffers (there: `MPI_Reduce(SUM)` to rank 0, here an all-reduce so every rank holds the
result).  All arithmetic runs in the library's kernels through its C ABI.
"""
import numpy as np

from .sharding import crt_prime_owner, shard_range


def crt_multiply_distributed(lib, a, b, p, dist, device, stream=None):
    """a, b: uint64 arrays of shape [n] or [batch, n], identical on every rank.
    lib: NttLibrary; dist: torch.distributed (initialised); device: torch device of this rank.
    Returns the products, shape [batch, 2n-1] (or [2n-1]), on every rank."""
    import torch
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    single = a.ndim == 1
    if single:
        a, b = a[None, :], b[None, :]
    batch, n = a.shape
    L = 1 << max(1, (2 * n - 2).bit_length())
    world, rank = dist.get_world_size(), dist.get_rank()
    ta = torch.from_numpy(a.view(np.int64)).to(device)
    tb = torch.from_numpy(b.view(np.int64)).to(device)
    res = torch.zeros((4, batch, L), dtype=torch.int32, device=device)
    st = stream if stream is not None else (torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None)
    for i in range(4):
        if crt_prime_owner(i, world) == rank:
            lib.check(lib.lib.ntt_crt_residue_product_device(ta.data_ptr(), tb.data_ptr(), res[i].data_ptr(), n, i, batch,
                                                             n, n, L, None, 0, st))
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    for i in range(4):
        dist.broadcast(res[i], src=crt_prime_owner(i, world))
    total = batch * L
    lo, hi = shard_range(total, rank, world)
    out = torch.zeros(total, dtype=torch.int64, device=device)
    if hi > lo:
        lib.check(lib.lib.ntt_crt_merge_device(res.data_ptr() + 4 * lo, total, out.data_ptr() + 8 * lo, hi - lo, int(p), st))
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.all_reduce(out)  # disjoint slices, zero elsewhere
    full = out.cpu().numpy().view(np.uint64).reshape(batch, L)[:, :2 * n - 1].copy()
    return full[0] if single else full
