"""Large-modulus products with the CRT work spread over ranks (one process per GPU).

What the reference's MPI variant does (main_mpi_CRT.cc:464-529): rank r runs the residue products of primes
i = r, r + size, ... (:470); every prime's residue vector is broadcast to all ranks (:498-501); every rank merges a contiguous
coefficient slice (:507-522); the zero-padded slices are summed onto rank 0 (`MPI_Reduce`, :525).  With four primes that
leaves ranks 4.. idle and moves every residue to every rank.

Here the unit of work is (prime, row) of a batch -- 4 * batch independent residue products -- dealt to the ranks in
contiguous runs, so that eight ranks are all busy from two rows on.  The exchange is ONE all-to-all: the (row, coefficient)
space is cut into `world` equal slices, and each rank sends, of every residue row it computed, only the part that lies in the
destination's slice (a 1/world share of the data per link, instead of a broadcast of everything).  Each rank merges its
slice of all four residues (`ntt_crt_merge_device`), and the slices are gathered -- to rank 0, like the reference, or to all
ranks on request.  No zero-padded all-reduce.  All arithmetic runs in the library's kernels through its C ABI; torch provides
device memory and the collectives (RCCL on GPUs, gloo on CPU in the tests); on a GPU everything is ordered on the current
stream, without host synchronisation between the steps.
"""
import numpy as np

from .sharding import crt_unit_range


def crt_multiply_distributed(lib, a, b, p, dist, device, stream=None, all_ranks=True):
    """a, b: uint64 arrays of shape [n] or [batch, n], identical on every rank (the reference broadcasts them,
    main_mpi_CRT.cc:569-572).  lib: NttLibrary; dist: torch.distributed (initialised); device: torch device of this rank.
    Returns the products, shape [batch, 2n-1] (or [2n-1]), on every rank (all_ranks=True) or on rank 0 only (None elsewhere)."""
    import torch
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    single = a.ndim == 1
    if single:
        a, b = a[None, :], b[None, :]
    batch, n = a.shape
    L = 1 << max(1, (2 * n - 2).bit_length())
    world, rank = dist.get_world_size(), dist.get_rank()
    st = stream if stream is not None else (torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None)
    ta = torch.from_numpy(a.view(np.int64)).to(device)
    tb = torch.from_numpy(b.view(np.int64)).to(device)

    # 1. residue products of this rank's (prime, row) units; unit u = prime * batch + row
    u_lo, u_hi = crt_unit_range(batch, rank, world)
    local = torch.zeros((max(u_hi - u_lo, 1), L), dtype=torch.int32, device=device)
    u = u_lo
    while u < u_hi:                       # one library call per prime: its rows in this rank's run are contiguous
        prime, row = divmod(u, batch)
        rows = min(u_hi, (prime + 1) * batch) - u
        lib.check(lib.lib.ntt_crt_residue_product_device(ta.data_ptr() + 8 * row * n, tb.data_ptr() + 8 * row * n,
                                                         local[u - u_lo].data_ptr(), n, prime, rows, n, n, L, None, 0, st))
        u += rows

    # 2. one all-to-all: slice s of the flattened (row, coefficient) space [batch * L] belongs to rank s
    total = batch * L
    per = -(-total // world)
    def slice_of(r):
        return min(r * per, total), min((r + 1) * per, total)
    def segments(units, sl):
        """For a run of units and a slice: (unit, start, stop) of the part of each unit's row that lies in the slice."""
        lo, hi = sl
        out = []
        for uu in range(*units):
            row = uu % batch
            s, e = max(lo, row * L), min(hi, (row + 1) * L)
            if s < e:
                out.append((uu, s, e))
        return out
    send_parts, send_counts = [], []
    for dst in range(world):
        segs = segments((u_lo, u_hi), slice_of(dst))
        send_counts.append(sum(e - s for _, s, e in segs))
        send_parts += [local[uu - u_lo, s - (uu % batch) * L:e - (uu % batch) * L] for uu, s, e in segs]
    send = torch.cat(send_parts) if send_parts else torch.zeros(0, dtype=torch.int32, device=device)
    my_lo, my_hi = slice_of(rank)
    recv_segs = [segments(crt_unit_range(batch, src, world), (my_lo, my_hi)) for src in range(world)]
    recv_counts = [sum(e - s for _, s, e in segs) for segs in recv_segs]
    recv = torch.empty(sum(recv_counts), dtype=torch.int32, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts)
    res = torch.zeros((4, per), dtype=torch.int32, device=device)      # this rank's slice of the four residue planes
    off = 0
    for segs in recv_segs:
        for uu, s, e in segs:
            res[uu // batch, s - my_lo:e - my_lo] = recv[off:off + e - s]
            off += e - s

    # 3. merge the slice, 4. gather the slices
    out = torch.zeros(per, dtype=torch.int64, device=device)
    if my_hi > my_lo:
        lib.check(lib.lib.ntt_crt_merge_device(res.data_ptr(), per, out.data_ptr(), my_hi - my_lo, int(p), st))
    if all_ranks:
        parts = [torch.empty_like(out) for _ in range(world)]
        dist.all_gather(parts, out)
    else:
        parts = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
        dist.gather(out, parts, dst=0)
        if rank != 0:
            return None
    full = torch.cat(parts)[:total].cpu().numpy().view(np.uint64).reshape(batch, L)[:, :2 * n - 1].copy()
    return full[0] if single else full
