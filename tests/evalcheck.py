"""Size-independent checks of long products: modular polynomial evaluation with torch integer tensors only (no call into
the library under test), for moduli below 2^55.  Test infrastructure.

A(x) mod p for a coefficient tensor is computed block-wise: the powers x^0..x^(B-1) are built by doubling with a 64-bit
modular multiply that takes the quotient estimate from float64 and the remainder from wrapping int64 arithmetic (exact,
because the estimate is off by at most a few units); block sums are accumulated in two half-words so they cannot overflow,
and the blocks are combined with Python integers on the host."""
import torch


def mulmod(a, b, p):
    """(a * b) mod p element-wise for int64 tensors with 0 <= a, b < p < 2^55 (b may be a Python int; beyond that the float64 quotient estimate is too coarse)."""
    if not torch.is_tensor(b):
        b = torch.tensor(int(b), dtype=torch.int64, device=a.device)
    q = (a.to(torch.float64) * b.to(torch.float64) / float(p)).floor().to(torch.int64)
    r = a * b - q * p            # wraps modulo 2^64; the true value lies within a few p of [0, p)
    return torch.remainder(r, p)


def powers(x, count, p, device):
    """x^0 .. x^(count-1) mod p as an int64 tensor (count a power of two)."""
    pw = torch.ones(1, dtype=torch.int64, device=device)
    step = int(x) % p
    while pw.numel() < count:
        pw = torch.cat([pw, mulmod(pw, step, p)])
        step = step * step % p
    return pw[:count]


def evaluate(coeffs, x, p, block=1 << 20, rows_per_pass=64):
    """sum_i coeffs[i] * x^i mod p (Python int).  coeffs: 1-D int64 tensor of canonical residues, any length."""
    dev = coeffs.device
    n = coeffs.numel()
    blk = min(block, 1 << max(0, (n - 1).bit_length()))
    pw = powers(x, blk, p, dev)
    full = n // blk
    sums = []
    for lo in range(0, full, rows_per_pass):
        hi = min(full, lo + rows_per_pass)
        t = mulmod(coeffs[lo * blk:hi * blk].view(hi - lo, blk), pw[None, :], p)
        lo_half = (t & 0x3FFFFFFF).sum(dim=1); hi_half = (t >> 30).sum(dim=1)
        sums += [int(l) + (int(h) << 30) for l, h in zip(lo_half.tolist(), hi_half.tolist())]
    if full * blk < n:
        tail = coeffs[full * blk:]
        t = mulmod(tail, pw[:tail.numel()], p)
        sums.append(int((t & 0x3FFFFFFF).sum()) + (int((t >> 30).sum()) << 30))
    xb = pow(int(x), blk, p)
    acc, scale = 0, 1
    for s in sums:
        acc = (acc + s % p * scale) % p
        scale = scale * xb % p
    return acc


def evaluate_rows(coeffs, x, p, rows_per_pass=8):
    """[A_r(x) mod p for every row r] of a 2-D int64 tensor of canonical residues (Python ints)."""
    dev = coeffs.device
    rows, n = coeffs.shape
    pw = powers(x, 1 << max(0, (n - 1).bit_length()), p, dev)[:n]
    out = []
    for lo in range(0, rows, rows_per_pass):
        t = mulmod(coeffs[lo:lo + rows_per_pass], pw[None, :], p)
        lo_half = (t & 0x3FFFFFFF).sum(dim=1); hi_half = (t >> 30).sum(dim=1)
        out += [(int(l) + (int(h) << 30)) % p for l, h in zip(lo_half.tolist(), hi_half.tolist())]
    return out


def evaluate_slab(slab, col0, C, x, p, rows=None, elems_per_pass=1 << 24):
    """This slab's share of A(x) for a coefficient vector laid out as an R x C matrix (element i = j + C*r at row r, column j):
    sum over the slab's rows r < rows and columns j = col0 .. col0 + Cl - 1 of slab[r][j - col0] * x^(j + C*r) mod p."""
    dev = slab.device
    R, Cl = slab.shape
    rows = R if rows is None else min(rows, R)
    pw = powers(x, 1 << max(0, (Cl - 1).bit_length()), p, dev)[:Cl]
    pw = mulmod(pw, pow(int(x), int(col0), p), p)
    per = max(1, elems_per_pass // Cl)
    row_vals = []
    for lo in range(0, rows, per):
        t = mulmod(slab[lo:min(rows, lo + per)], pw[None, :], p)
        lo_half = (t & 0x3FFFFFFF).sum(dim=1); hi_half = (t >> 30).sum(dim=1)
        row_vals += [(int(l) + (int(h) << 30)) % p for l, h in zip(lo_half.tolist(), hi_half.tolist())]
    xc = pow(int(x), int(C), p)
    acc, scale = 0, 1
    for v in row_vals:
        acc = (acc + v * scale) % p
        scale = scale * xc % p
    return acc
