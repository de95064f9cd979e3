"""One very long polynomial product by a four-step transform, on one GPU or sharded over ranks.

The length-L transform (L = 2^l >= 2n-1) is viewed as an R x C matrix, element i = j + C*r:
  1. every rank holds a slab of columns [R][C/G] and transforms it along the row axis (length R),
  2. multiplies by the twiddles w_L^(j*k1),
  3. one all-to-all turns column slabs into row slabs [R/G][C],
  4. each row is transformed along the column axis (length C).
The spectra of both operands end up in the same (scrambled) order, the pointwise product is
local, and the inverse runs the mirror sequence: three all-to-alls per product in total.
This replaces the reference's distributed transform, which keeps the whole array on every rank
and all-reduces it after every butterfly stage (main_mpi_t.cc:164-222).

All arithmetic is done by the library's 64-bit-field kernels through its C ABI; torch is used
for device memory, the local re-tiling around the all-to-all and the collective itself
(RCCL on GPUs, gloo on CPU in the tests)."""
import numpy as np


def _log2(x):
    assert x > 0 and x & (x - 1) == 0, "power of two expected"
    return x.bit_length() - 1


def plan_shape(n, world):
    """(log_total, log_rows, log_cols) for operands of n coefficients on `world` ranks."""
    log_total = max(1, (2 * n - 2).bit_length()) if n > 1 else 1
    log_rows = log_total // 2
    g = _log2(world)
    log_rows = max(log_rows, g)
    if log_total - log_rows < g:
        raise ValueError("transform too short for this many ranks")
    return log_total, log_rows, log_total - log_rows


class FourStep:
    """Four-step product context of one rank.  lib: NttLibrary; dist: torch.distributed or None."""

    def __init__(self, lib, p, n, device, dist=None):
        import torch
        self.torch, self.lib, self.p, self.n, self.device, self.dist = torch, lib, int(p), int(n), device, dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.log_total, self.log_rows, self.log_cols = plan_shape(n, self.world)
        self.R, self.C = 1 << self.log_rows, 1 << self.log_cols
        self.Cl, self.Rl = self.C // self.world, self.R // self.world
        self.stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None

    # -- data placement ---------------------------------------------------------------
    def column_slab(self, full):
        """This rank's [R][C/G] slab of a length-n coefficient vector (numpy uint64), zero padded."""
        t = self.torch
        L = self.R * self.C
        v = np.zeros(L, dtype=np.uint64)
        v[: self.n] = full[: self.n]
        slab = v.reshape(self.R, self.C)[:, self.rank * self.Cl:(self.rank + 1) * self.Cl]
        return t.from_numpy(np.ascontiguousarray(slab).view(np.int64)).to(self.device)

    def _sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def _columns_to_rows(self, x):
        """[R][Cl] column slab -> [Rl][C] row slab (all-to-all + local re-tiling)."""
        t = self.torch
        if self.world == 1:
            return x
        send = x.contiguous()                       # rows g*Rl..(g+1)*Rl-1 are the chunk for rank g
        recv = t.empty_like(send)
        self._sync()
        self.dist.all_to_all_single(recv, send)
        # recv[g''] = [Rl][Cl] tile holding columns of rank g'' -> interleave into [Rl][G*Cl]
        return recv.view(self.world, self.Rl, self.Cl).permute(1, 0, 2).contiguous().view(self.Rl, self.C)

    def _rows_to_columns(self, z):
        t = self.torch
        if self.world == 1:
            return z
        send = z.view(self.Rl, self.world, self.Cl).permute(1, 0, 2).contiguous()
        recv = t.empty_like(send)
        self._sync()
        self.dist.all_to_all_single(recv, send)
        return recv.view(self.R, self.Cl)

    # -- transforms ---------------------------------------------------------------------
    def forward(self, slab):
        """Column slab (values any 64-bit) -> row slab of the spectrum, scrambled order."""
        L, lib, p, lr = self.lib.lib, self.lib, self.p, self.log_total
        x = slab
        lib.check(L.ntt64_load_device(x.data_ptr(), 0, x.numel(), x.data_ptr(), x.numel(), 1, p, lr, self.stream))
        lib.check(L.ntt64_transform_device(x.data_ptr(), p, lr, self.log_rows, self.Cl, 1, 0, self.stream))
        lib.check(L.ntt64_twiddle_device(x.data_ptr(), p, lr, self.R, self.Cl, 0, self.rank * self.Cl, self.log_rows, 1, 0, self.stream))
        z = self._columns_to_rows(x)
        lib.check(L.ntt64_transform_device(z.data_ptr(), p, lr, self.log_cols, 1, self.Rl, 0, self.stream))
        return z

    def inverse(self, z):
        """Row slab of a spectrum -> column slab of canonical coefficients [R][Cl] (c[j + C*r])."""
        L, lib, p, lr = self.lib.lib, self.lib, self.p, self.log_total
        lib.check(L.ntt64_transform_device(z.data_ptr(), p, lr, self.log_cols, 1, self.Rl, 1, self.stream))
        x = self._rows_to_columns(z)
        lib.check(L.ntt64_twiddle_device(x.data_ptr(), p, lr, self.R, self.Cl, 0, self.rank * self.Cl, self.log_rows, 1, 1, self.stream))
        lib.check(L.ntt64_transform_device(x.data_ptr(), p, lr, self.log_rows, self.Cl, 1, 1, self.stream))
        out = self.torch.empty_like(x)
        lib.check(L.ntt64_finalize_device(x.data_ptr(), x.numel(), out.data_ptr(), 0, x.numel(), 1, p, lr, self.log_total, self.stream))
        return out

    def multiply_slabs(self, a_slab, b_slab):
        """Column slabs of a and b -> column slab of the product."""
        za = self.forward(a_slab)
        zb = self.forward(b_slab)
        self.lib.check(self.lib.lib.ntt64_pointwise_device(za.data_ptr(), zb.data_ptr(), za.numel(), self.p, self.log_total, self.stream))
        return self.inverse(za)

    def gather(self, slab):
        """All ranks' column slabs -> the full product (numpy uint64, 2n-1 values) on every rank."""
        t = self.torch
        self._sync()
        if self.world > 1:
            parts = [t.empty_like(slab) for _ in range(self.world)]
            self.dist.all_gather(parts, slab.contiguous())
            full = t.cat(parts, dim=1)
        else:
            full = slab
        return full.cpu().numpy().view(np.uint64).reshape(-1)[: 2 * self.n - 1].copy()


def poly_multiply_fourstep(lib, a, b, p, device, dist=None):
    """Full product of two numpy uint64 coefficient vectors (identical on every rank)."""
    fs = FourStep(lib, p, len(a), device, dist)
    c = fs.multiply_slabs(fs.column_slab(np.asarray(a, dtype=np.uint64)), fs.column_slab(np.asarray(b, dtype=np.uint64)))
    return fs.gather(c)
