"""Large-modulus products with the CRT work spread over ranks (one process per GPU), device-resident.

What the reference's MPI variant does (main_mpi_CRT.cc:464-529): rank r runs the residue products of primes
i = r, r + size, ... (:470); every prime's residue vector is broadcast to all ranks (:498-501); every rank merges a contiguous
coefficient slice (:507-522); the zero-padded slices are summed onto rank 0 (`MPI_Reduce`, :525).  With four primes that
leaves ranks 4.. idle and moves every residue to every rank.

Here the unit of work is (prime, row) of a batch -- 4 * batch independent residue products -- dealt to the ranks in
contiguous runs, so that eight ranks are all busy from two rows on.  The exchange is ONE all-to-all: the coefficient axis
is cut into `world` equal slices of Lw = L / world coefficients, and each rank sends, of every residue row it computed,
only the slice the destination owns (a 1/world share of the data per link, instead of a broadcast of everything).  The
send buffer is the rank's residue rows viewed as [units][world][Lw] with the first two axes swapped -- one strided copy,
no per-segment packing -- and because the ranks hold contiguous runs of units in rank order, what arrives is already the
four residue planes [4][batch][Lw] of the receiver's slice: nothing is unpacked.  Each rank merges its slice
(`ntt_crt_merge_device`) and keeps it; gathering the slices is optional.  No zero-padded all-reduce, no host memory
between the residue products and the merge.

All arithmetic runs in the library's kernels through its C ABI; torch provides device memory and the collectives (RCCL on
GPUs, gloo on CPU in the tests).  On a GPU with RCCL everything is ordered on torch's current stream and the host never
waits between the steps; with any other backend on a GPU (rehearsals over gloo) the exchange goes through host memory,
bracketed by device synchronisation."""
import numpy as np

from .sharding import crt_unit_range


def collectives_follow_the_stream(dist, device):
    """True when this process group's transfers are enqueued on the device stream (RCCL), or there is no device stream
    to race with (CPU tensors)."""
    return device.type != "cuda" or dist is None or dist.get_backend() == "nccl"


class CrtDistributed:
    """Context of one rank for batches of `batch` products of n coefficients modulo p, the same operands on every rank
    (the reference broadcasts them, main_mpi_CRT.cc:569-572).  Buffers are allocated once, here.

    multiply(ta, tb) -> this rank's slice of the products: int64 tensor [batch][Lw], coefficients
    [rank * Lw, (rank + 1) * Lw) of every row (positions from 2n-1 on are padding and hold zeros).
    gather(slice) -> the full products [batch][2n-1] on every rank or on rank 0."""

    def __init__(self, lib, p, n, batch, device, dist=None):
        import torch
        self.torch, self.lib, self.p, self.n, self.batch, self.device = torch, lib, int(p), int(n), int(batch), device
        self.dist = dist if dist is not None and dist.is_initialized() and dist.get_world_size() > 1 else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.L = 1 << max(1, (2 * self.n - 2).bit_length())
        self.Lw = -(-self.L // self.world)                      # coefficients per slice
        self.Lp = self.Lw * self.world                          # row length of the residue buffer (>= L)
        self.u_lo, self.u_hi = crt_unit_range(self.batch, self.rank, self.world)
        self.units = [crt_unit_range(self.batch, r, self.world) for r in range(self.world)]
        nu = self.u_hi - self.u_lo
        # residue rows of this rank's units; the products write 2n-1 values per row, the rest stays zero
        self.local = torch.zeros((max(nu, 1), self.Lp), dtype=torch.int32, device=device)
        self.send = torch.empty((self.world, nu, self.Lw), dtype=torch.int32, device=device)
        self.recv = torch.empty((4 * self.batch, self.Lw), dtype=torch.int32, device=device)
        self.out = torch.empty((self.batch, self.Lw), dtype=torch.int64, device=device)
        self.equal_split = all(hi - lo == nu for lo, hi in self.units)
        self.in_splits = [nu * self.Lw] * self.world
        self.out_splits = [(hi - lo) * self.Lw for lo, hi in self.units]
        self.stream_ordered = collectives_follow_the_stream(self.dist, device)
        ws_rows = min(self.batch, max(nu, 1))
        self.ws_bytes = lib.lib.ntt_workspace_bytes(self.n, int(lib.lib.ntt_crt_prime(0)), ws_rows)
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=device)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    def _host_wait(self):
        if not self.stream_ordered:
            self.torch.cuda.synchronize(self.device)

    def multiply(self, ta, tb):
        """ta, tb: int64 device tensors [batch][n] (any 64-bit words; they are reduced modulo each prime on the fly)."""
        L, lib, n, st = self.lib.lib, self.lib, self.n, self._stream()
        assert ta.shape == (self.batch, n) and tb.shape == (self.batch, n) and ta.is_contiguous() and tb.is_contiguous()
        # 1. residue products of this rank's (prime, row) units; unit u = prime * batch + row.  One library call per
        #    prime: its rows inside this rank's run are contiguous (at most four calls)
        u = self.u_lo
        while u < self.u_hi:
            prime, row = divmod(u, self.batch)
            rows = min(self.u_hi, (prime + 1) * self.batch) - u
            lib.check(L.ntt_crt_residue_product_device(ta.data_ptr() + 8 * row * n, tb.data_ptr() + 8 * row * n,
                                                       self.local[u - self.u_lo].data_ptr(), n, prime, rows, n, n, self.Lp,
                                                       self.ws.data_ptr(), self.ws_bytes, st))
            u += rows
        # 2. one all-to-all of coefficient slices
        nu = self.u_hi - self.u_lo
        if self.world == 1:
            recv = self.local[:nu]
        else:
            if nu:
                self.send.copy_(self.local[:nu].view(nu, self.world, self.Lw).transpose(0, 1))
            splits = {} if self.equal_split else {"output_split_sizes": self.out_splits, "input_split_sizes": self.in_splits}
            if self.stream_ordered:
                self.dist.all_to_all_single(self.recv.view(-1), self.send.view(-1), **splits)
            else:                                    # through host memory, bracketed by device synchronisation
                host_in = self.torch.empty(self.recv.numel(), dtype=self.torch.int32)
                self.dist.all_to_all_single(host_in, self.send.view(-1).cpu(), **splits)
                self.recv.view(-1).copy_(host_in)
                self._host_wait()
            recv = self.recv
        # 3. merge this rank's slice of the four residue planes [4][batch][Lw]
        count = self.batch * self.Lw
        lib.check(L.ntt_crt_merge_device(recv.data_ptr(), count, self.out.data_ptr(), count, self.p, st))
        return self.out

    def gather(self, out, all_ranks=True):
        """Slices of all ranks -> full products, int64 tensor [batch][2n-1] (None on ranks other than 0 unless all_ranks)."""
        t = self.torch
        if self.world == 1:
            return out[:, :2 * self.n - 1]
        if not self.stream_ordered:                  # through host memory
            host = out.cpu()
            parts = [t.empty_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host)
            if not all_ranks and self.rank != 0:
                return None
            return t.stack(parts, dim=1).reshape(self.batch, self.Lp)[:, :2 * self.n - 1].to(self.device)
        if all_ranks:
            parts = [t.empty_like(out) for _ in range(self.world)]
            self.dist.all_gather(parts, out)
        else:
            parts = [t.empty_like(out) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(out, parts, dst=0)
            if self.rank != 0:
                return None
        return t.stack(parts, dim=1).reshape(self.batch, self.Lp)[:, :2 * self.n - 1]


def crt_multiply_distributed(lib, a, b, p, dist, device, all_ranks=True):
    """Host-array face of CrtDistributed (tests, small calls): a, b uint64 arrays of shape [n] or [batch, n], identical on
    every rank.  Returns the products, shape [batch, 2n-1] (or [2n-1]), on every rank (all_ranks=True) or on rank 0 only
    (None elsewhere)."""
    import torch
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    single = a.ndim == 1
    if single:
        a, b = a[None, :], b[None, :]
    batch, n = a.shape
    ctx = CrtDistributed(lib, p, n, batch, device, dist)
    ta = torch.from_numpy(a.view(np.int64)).to(device)
    tb = torch.from_numpy(b.view(np.int64)).to(device)
    full = ctx.gather(ctx.multiply(ta, tb), all_ranks=all_ranks)
    if full is None:
        return None
    full = full.cpu().numpy().view(np.uint64).copy()
    return full[0] if single else full
