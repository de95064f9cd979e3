"""One very long polynomial product by a four-step transform, on one GPU or sharded over ranks.

The length-L transform (L = 2^l >= 2n-1) is viewed as an R x C matrix, element i = j + C*r:
  1. every rank holds a slab of columns [R][C/G] and runs the first log2 R levels of the long transform on it
     (`ntt64_fourstep_rows_device`: the transforms along the rows axis together with the twiddles w_L^(j*k1)
     that join the two axes, in the same passes over memory);
  2. ONE all-to-all turns column slabs into row slabs.  The send side needs no packing (rows g*R/G.. of the slab
     are the chunk for rank g) and the receive side no re-tiling: the remaining levels run directly on the
     [G][R/G][C/G] layout the exchange leaves (`ntt64_fourstep_cols_device`);
  3. the exchange is issued in `chunks` pieces of rows, so that the column-axis transforms of a piece run while
     the later pieces are still on the links.
The spectra of both operands end up in the same (scrambled) order, the pointwise product is local, and the
inverse runs the mirror sequence: three all-to-alls per product in total, each moving (G-1)/G of the rank's
slab.  Everything is ordered on the device stream; the host never waits between the steps.
A product (`multiply_slabs`) folds the element-wise ends into the passes next to them: the operands are reduced and
zero-padded by the first row pass (`ntt64_fourstep_rows_load_device`, the padding rows are never read), the column
passes of both operands, the pointwise product and the inverse column passes of a piece are one call
(`ntt64_fourstep_cols_product_device`: their last forward levels, the product and the first inverse levels one
kernel), and the last row pass writes the scaled canonical coefficients (`ntt64_fourstep_rows_store_device`).
The matrix shape is chosen for the fewest passes over memory (`plan_shape`).
This replaces the reference's distributed transform, which keeps the whole array on every rank and all-reduces
it after every butterfly stage (main_mpi_t.cc:164-222).

All arithmetic is done by the library's 64-bit-field kernels through its C ABI; torch provides device memory
and the collective (RCCL on GPUs, gloo on CPU in the tests)."""
import numpy as np


def _log2(x):
    assert x > 0 and x & (x - 1) == 0, "power of two expected"
    return x.bit_length() - 1


def _passes(log_rows, log_cols_local, log_parts):
    """Passes over memory of one transform for a matrix shape: the row axis goes through LDS tiles of up to 2^9 rows
    (a register pass when it has fewer than 2^5), the column axis ends in one pass over 4096-word chunks with tiled
    passes of up to 9 levels in front, plus a register pass over the parts an exchange leaves."""
    rows = 1 if log_rows <= 4 else -(-log_rows // 9)
    cols = 1 if log_cols_local <= 12 else 1 + -(-(log_cols_local - 12) // 9)
    return rows + cols + (1 if log_parts else 0)


def plan_shape(n, world):
    """(log_total, log_rows, log_cols) for operands of n coefficients on `world` ranks: of the shapes with the fewest
    passes over memory per transform the one closest to square (the exchange then moves the squarest pieces)."""
    log_total = max(1, (2 * n - 2).bit_length()) if n > 1 else 1
    g = _log2(world)
    if log_total < 2 * g:
        raise ValueError("transform too short for this many ranks")
    best = None
    for log_rows in range(g, log_total - g + 1):
        cost = (_passes(log_rows, log_total - log_rows - g, g), abs(2 * log_rows - log_total), log_rows)
        if best is None or cost < best[0]:
            best = (cost, log_rows)
    log_rows = best[1]
    return log_total, log_rows, log_total - log_rows


class _HostStagedExchange:
    """The exchange of one row piece for process groups that cannot take device memory on the device stream (gloo with GPU
    tensors: rehearsals of the rank logic on a box with fewer GPUs than ranks): the chunks travel through host tensors,
    bracketed by device synchronisation.  Same interface as a pending transfer: wait()."""

    def __init__(self, fs, send, recv, lo, hi):
        t, d = fs.torch, fs.dist
        t.cuda.synchronize(fs.device)                    # the kernels that produced `send` have finished
        recv[fs.rank, lo:hi].copy_(send[fs.rank, lo:hi])
        self.fs, self.recv, self.lo, self.hi = fs, recv, lo, hi
        self.host_out = {g: send[g, lo:hi].cpu() for g in range(fs.world) if g != fs.rank}
        self.host_in = {g: t.empty_like(v) for g, v in self.host_out.items()}
        ops = []
        for g in self.host_out:
            ops.append(d.P2POp(d.irecv, self.host_in[g], g))
            ops.append(d.P2POp(d.isend, self.host_out[g], g))
        self.works = d.batch_isend_irecv(ops) if ops else []

    def wait(self):
        for w in self.works:
            w.wait()
        for g, v in self.host_in.items():
            self.recv[g, self.lo:self.hi].copy_(v)
        self.fs.torch.cuda.synchronize(self.fs.device)


class FourStep:
    """Four-step product context of one rank.  lib: NttLibrary; dist: torch.distributed or None."""

    def __init__(self, lib, p, n, device, dist=None, chunks=4):
        import torch
        self.torch, self.lib, self.p, self.n, self.device, self.dist = torch, lib, int(p), int(n), device, dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        if self.world > 16:
            raise ValueError("the split layout of the column pass supports at most 16 ranks")
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.log_total, self.log_rows, self.log_cols = plan_shape(n, self.world)
        self.R, self.C = 1 << self.log_rows, 1 << self.log_cols
        self.Cl, self.Rl = self.C // self.world, self.R // self.world
        self.log_parts = _log2(self.world)
        self.chunks = max(1, min(int(chunks), self.Rl)) if self.world > 1 else 1
        while self.Rl % self.chunks:
            self.chunks -= 1
        self.stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None
        # RCCL enqueues its transfers on the device stream, which is what orders them against the library's kernels; any
        # other backend on a GPU (gloo) moves device memory from the host side, so the exchanges are then bracketed by
        # device synchronisation instead
        from .distributed import collectives_follow_the_stream
        self.stream_ordered = collectives_follow_the_stream(dist if self.world > 1 else None, device)

    # -- data placement ---------------------------------------------------------------
    def column_slab(self, full):
        """This rank's [R][C/G] slab of a length-n coefficient vector (numpy uint64), zero padded."""
        t = self.torch
        L = self.R * self.C
        v = np.zeros(L, dtype=np.uint64)
        v[: self.n] = full[: self.n]
        slab = v.reshape(self.R, self.C)[:, self.rank * self.Cl:(self.rank + 1) * self.Cl]
        return t.from_numpy(np.ascontiguousarray(slab).view(np.int64)).to(self.device)

    def column_slab_device(self, full):
        """The same from a coefficient vector that is already on the device (int64 tensor of n words): only the rows
        that hold coefficients are filled; the rest of the slab is never read by multiply_slabs."""
        t = self.torch
        slab = t.empty((self.R, self.Cl), dtype=t.int64, device=self.device)
        lo, hi = self.rank * self.Cl, (self.rank + 1) * self.Cl
        whole = self.n // self.C                                  # complete matrix rows of coefficients
        if whole:
            slab[:whole].copy_(full[: whole * self.C].view(whole, self.C)[:, lo:hi])
        rest = self.n - whole * self.C
        if rest and whole < self.R:
            slab[whole].zero_()
            if rest > lo:
                slab[whole, : min(rest, hi) - lo].copy_(full[whole * self.C + lo: whole * self.C + min(rest, hi)])
        return slab

    def _exchange(self, send, recv, rows):
        """All-to-all of rows [rows[0], rows[1]) of every rank's chunk: send / recv are [G][Rl][Cl] arrays, chunk g of
        `send` goes to rank g and chunk g of `recv` comes from it.  Written as one batch of point-to-point transfers
        (what an all-to-all is on RCCL: a group of sends and receives over all xGMI links at once), so that the pieces
        need no packing and the same code runs on gloo.  Returns the pending transfers."""
        lo, hi = rows
        if not self.stream_ordered:
            return [_HostStagedExchange(self, send, recv, lo, hi)]
        recv[self.rank, lo:hi].copy_(send[self.rank, lo:hi])
        ops = []
        for g in range(self.world):
            if g != self.rank:
                ops.append(self.dist.P2POp(self.dist.irecv, recv[g, lo:hi], g))
                ops.append(self.dist.P2POp(self.dist.isend, send[g, lo:hi], g))
        return self.dist.batch_isend_irecv(ops) if ops else []

    def _wait(self, works):
        """Completion of pending transfers: ordered on the stream with RCCL, waited for by the host otherwise."""
        for w in works:
            w.wait()

    def _row_pieces(self):
        per = self.Rl // self.chunks
        return [(k * per, (k + 1) * per) for k in range(self.chunks)]

    # -- transforms ---------------------------------------------------------------------
    def forward(self, slab):
        """Column slab [R][Cl] (any 64-bit values) -> spectrum in the received layout [G][Rl][Cl], scrambled order."""
        L, lib, p, lt = self.lib.lib, self.lib, self.p, self.log_total
        x = slab
        lib.check(L.ntt64_load_device(x.data_ptr(), 0, x.numel(), x.data_ptr(), x.numel(), 1, p, lt, self.stream))
        lib.check(L.ntt64_fourstep_rows_device(x.data_ptr(), p, lt, self.log_rows, self.Cl, self.rank * self.Cl, 0, self.stream))
        if self.world == 1:
            lib.check(L.ntt64_fourstep_cols_device(x.data_ptr(), p, lt, self.log_cols, 0, 0, self.R, 0, self.stream))
            return x
        send = x.view(self.world, self.Rl, self.Cl)
        recv = self.torch.empty_like(send)
        pieces = self._row_pieces()
        works = [self._exchange(send, recv, rows) for rows in pieces]          # all issued behind the row pass
        for pending, (lo, hi) in zip(works, pieces):
            self._wait(pending)                                                 # orders the stream, not the host, with RCCL
            lib.check(L.ntt64_fourstep_cols_device(recv.data_ptr() + 8 * lo * self.Cl, p, lt, self.log_cols, self.log_parts,
                                                   self.Rl * self.Cl, hi - lo, 0, self.stream))
        return recv

    def inverse(self, z):
        """Spectrum in the received layout -> column slab of canonical coefficients [R][Cl] (c[j + C*r])."""
        L, lib, p, lt = self.lib.lib, self.lib, self.p, self.log_total
        if self.world == 1:
            lib.check(L.ntt64_fourstep_cols_device(z.data_ptr(), p, lt, self.log_cols, 0, 0, self.R, 1, self.stream))
            x = z.view(self.R, self.Cl)
        else:
            z = z.view(self.world, self.Rl, self.Cl)
            back = self.torch.empty_like(z)
            works = []
            for lo, hi in self._row_pieces():
                lib.check(L.ntt64_fourstep_cols_device(z.data_ptr() + 8 * lo * self.Cl, p, lt, self.log_cols, self.log_parts,
                                                       self.Rl * self.Cl, hi - lo, 1, self.stream))
                works.append(self._exchange(z, back, (lo, hi)))                # travels while the next piece is transformed
            for pending in works:
                self._wait(pending)
            x = back.view(self.R, self.Cl)
        lib.check(L.ntt64_fourstep_rows_device(x.data_ptr(), p, lt, self.log_rows, self.Cl, self.rank * self.Cl, 1, self.stream))
        out = self.torch.empty_like(x)
        lib.check(L.ntt64_finalize_device(x.data_ptr(), x.numel(), out.data_ptr(), 0, x.numel(), 1, p, lt, self.log_total, self.stream))
        return out

    def multiply_slabs(self, a_slab, b_slab, rows_valid=None):
        """Column slabs of a and b (any 64-bit words; rows from rows_valid on are taken as zero) -> column slab of the
        product, canonical.  The slabs are overwritten."""
        L, lib, p, lt, t = self.lib.lib, self.lib, self.p, self.log_total, self.torch
        if rows_valid is None:
            rows_valid = min(self.R, -(-self.n // self.C))       # rows of the matrix that hold coefficients
        spectra = []
        pending = []
        pieces = self._row_pieces()
        for slab in (a_slab, b_slab):
            lib.check(L.ntt64_fourstep_rows_load_device(slab.data_ptr(), rows_valid, slab.data_ptr(), p, lt, self.log_rows, self.Cl,
                                                        self.rank * self.Cl, self.stream))
            if self.world == 1:
                spectra.append(slab)
                continue
            send = slab.view(self.world, self.Rl, self.Cl)
            recv = t.empty_like(send)
            pending.append([self._exchange(send, recv, rows) for rows in pieces])
            spectra.append(recv)
        za, zb = spectra
        if self.world == 1:
            lib.check(L.ntt64_fourstep_cols_product_device(za.data_ptr(), zb.data_ptr(), p, lt, self.log_cols, 0, 0, self.R, self.stream))
            x = za.view(self.R, self.Cl)
        else:
            back = t.empty_like(za)
            returning = []
            for k, (lo, hi) in enumerate(pieces):
                self._wait(pending[0][k] + pending[1][k])            # orders the stream, not the host, with RCCL
                off = 8 * lo * self.Cl
                lib.check(L.ntt64_fourstep_cols_product_device(za.data_ptr() + off, zb.data_ptr() + off, p, lt, self.log_cols,
                                                               self.log_parts, self.Rl * self.Cl, hi - lo, self.stream))
                returning.append(self._exchange(za, back, (lo, hi)))   # travels while the next piece is worked on
            for works in returning:
                self._wait(works)
            x = back.view(self.R, self.Cl)
        out = t.empty_like(x)
        lib.check(L.ntt64_fourstep_rows_store_device(x.data_ptr(), out.data_ptr(), p, lt, self.log_rows, self.Cl, self.rank * self.Cl,
                                                     self.stream))
        return out

    def multiply_slabs_stepwise(self, a_slab, b_slab):
        """The same product from the separate transforms (forward, forward, pointwise, inverse): the form to use when the
        spectra themselves are wanted."""
        za = self.forward(a_slab)
        zb = self.forward(b_slab)
        self.lib.check(self.lib.lib.ntt64_pointwise_device(za.data_ptr(), zb.data_ptr(), za.numel(), self.p, self.log_total, self.stream))
        return self.inverse(za)

    def gather(self, slab, all_ranks=False):
        """The ranks' column slabs -> the full product (numpy uint64, 2n-1 values) on rank 0 (None elsewhere), or on every
        rank with all_ranks=True.  A product this long normally stays sharded; this is for checking."""
        full = self.gather_device(slab, all_ranks)
        return None if full is None else full.cpu().numpy().view(np.uint64).copy()

    def gather_device(self, slab, all_ranks=False):
        """The same, staying on the device: int64 tensor of 2n-1 words."""
        t = self.torch
        slab = slab.contiguous()
        if self.world > 1 and not self.stream_ordered:       # through host memory (see _HostStagedExchange)
            host = slab.cpu()
            parts = [t.empty_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host)
            if not all_ranks and self.rank != 0:
                return None
            return t.cat(parts, dim=1).to(self.device).reshape(-1)[: 2 * self.n - 1]
        if self.world > 1:
            if all_ranks:
                parts = [t.empty_like(slab) for _ in range(self.world)]
                self.dist.all_gather(parts, slab)
            else:
                parts = [t.empty_like(slab) for _ in range(self.world)] if self.rank == 0 else None
                self.dist.gather(slab, parts, dst=0)
                if self.rank != 0:
                    return None
            full = t.cat(parts, dim=1)
        else:
            full = slab
        return full.reshape(-1)[: 2 * self.n - 1]


def poly_multiply_fourstep(lib, a, b, p, device, dist=None, chunks=4):
    """Full product of two numpy uint64 coefficient vectors (identical on every rank), returned on every rank."""
    fs = FourStep(lib, p, len(a), device, dist, chunks)
    c = fs.multiply_slabs(fs.column_slab(np.asarray(a, dtype=np.uint64)), fs.column_slab(np.asarray(b, dtype=np.uint64)))
    return fs.gather(c, all_ranks=True)
