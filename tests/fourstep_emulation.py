"""The sharded four-step product with all ranks emulated in ONE process on one device: every rank's row passes, the
exchanges done as tensor copies, every rank's column passes on the layout an exchange leaves.  It drives the same C
entry points in the same order as ntt_amd.fourstep.FourStep.multiply_slabs does on real ranks, so the kernels of the
multi-rank path (the pass over the parts, transforms on sub-ranges of a part) run on whatever device the library serves
-- in particular on a single GPU."""
import numpy as np

from ntt_amd.fourstep import plan_shape


def emulated_fourstep_product(lib, a, b, p, world, to_dev, from_dev, torch, stream, pieces=2, all_rows=False):
    """Product of two numpy uint64 vectors of equal length through `world` emulated ranks; returns 2n-1 values.
    all_rows: tell the loading pass that every row of the (zero-padded) slab holds data, which takes its general path
    instead of the one that skips the padding half."""
    L = lib.lib
    n = len(a)
    lt, lr, lc = plan_shape(n, world)
    R, C = 1 << lr, 1 << lc
    Cl, Rl = C // world, R // world
    g = world.bit_length() - 1
    rows_valid = R if all_rows else min(R, -(-n // C))

    def slabs(v):
        full = np.zeros(R * C, dtype=np.uint64); full[:n] = v
        m = full.reshape(R, C)
        return [to_dev(np.ascontiguousarray(m[:, r * Cl:(r + 1) * Cl])) for r in range(world)]

    sa, sb = slabs(a), slabs(b)
    for r in range(world):
        for s in (sa, sb):
            lib.check(L.ntt64_fourstep_rows_load_device(s[r].data_ptr(), rows_valid, s[r].data_ptr(), p, lt, lr, Cl, r * Cl, stream))

    def exchange(parts):          # received[r][g] = sent[g][r], chunks of [Rl][Cl]
        views = [x.view(world, Rl, Cl) for x in parts]
        return [torch.stack([views[src][r] for src in range(world)]).contiguous() for r in range(world)]

    za, zb = exchange(sa), exchange(sb)
    per = max(1, Rl // pieces)
    for r in range(world):
        for lo in range(0, Rl, per):      # sub-ranges of the rows of a part, as the overlapped exchange delivers them
            rows = min(per, Rl - lo)
            off = 8 * lo * Cl
            lib.check(L.ntt64_fourstep_cols_product_device(za[r].data_ptr() + off, zb[r].data_ptr() + off, p, lt, lc, g, Rl * Cl, rows, stream))
    back = exchange(za)
    outs = []
    for r in range(world):
        x = back[r].view(R, Cl)
        out = torch.empty_like(x)
        lib.check(L.ntt64_fourstep_rows_store_device(x.data_ptr(), out.data_ptr(), p, lt, lr, Cl, r * Cl, stream))
        outs.append(out)
    full = torch.cat(outs, dim=1)
    return from_dev(full).reshape(-1)[: 2 * n - 1].copy()
