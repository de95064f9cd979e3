"""The benchmark configurations at their full sizes, through the C ABI on one GPU.

 * 65536 products of n = 4096 in one call (one workgroup per product), for p = 7340033 with the reference's filetest/0-3
   in rows 0-3, and for p = 998244353: sampled rows against the oracle bit for bit, every row through the evaluation
   identity A(x) B(x) = C(x) at random points, and linearity in the first operand on all rows.
 * 64 large-modulus products (n = 2^17, p = 19 * 2^46 + 1, four-prime CRT route of main_naive.cc:178-230): every row
   against the oracle.
 * One product of n = 2^28 coefficients modulo the same prime in the 64-bit field (what replaces the distributed
   transform of main_mpi_t.cc:139-291): no CPU product exists at this size, so it is checked through properties --
   shift by a monomial, the all-ones product, and evaluations at x = 1, x = -1 and two random points."""
import numpy as np
import pytest

from evalcheck import evaluate

pytestmark = pytest.mark.gpu
BIGP = 1337006139375617


def _eval_rows(torch, t, count, x, p):
    """sum_i t[r, i] x^i mod p for every row r (p < 2^31), int64 tensor of shape [rows]."""
    dev = t.device
    pw = torch.ones(1, dtype=torch.int64, device=dev)
    step = x % p
    while pw.numel() < count:
        pw = torch.cat([pw, pw * step % p])
        step = step * step % p
    pw = pw[:count]
    acc = torch.empty(t.shape[0], dtype=torch.int64, device=dev)
    for lo in range(0, t.shape[0], 4096):
        v = t[lo:lo + 4096, :count].to(torch.int64) & 0xFFFFFFFF
        acc[lo:lo + 4096] = (v * pw % p).sum(dim=1) % p
    return acc


@pytest.mark.parametrize("p", [7340033, 998244353])
def test_65536_products_of_n_4096(gpu_lib, oracle, p):
    import torch
    from golden_util import iter_golden
    L = gpu_lib.lib
    n, batch, n_out = 4096, 65536, 8191
    g = torch.Generator(device="cuda"); g.manual_seed(0xC0FFEE00 + p % 97)
    a = torch.randint(0, p, (batch, n), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, p, (batch, n), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    golden = {}
    if p == 7340033:
        for name, gn, gp, ga, gb, gout in iter_golden():
            if name.startswith("filetest"):
                row = int(name[len("filetest")])
                a[row] = torch.from_numpy(ga.astype(np.int32)).cuda(); b[row] = torch.from_numpy(gb.astype(np.int32)).cuda()
                golden[row] = gout
        assert sorted(golden) == [0, 1, 2, 3]
    out = torch.full((batch, n_out), -1, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    gpu_lib.check(L.ntt_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, batch, n, n, n_out, None, 0, st))
    torch.cuda.synchronize()
    assert int(out.view(torch.int32).min()) >= 0 and int(out.max()) < p          # canonical residues everywhere
    for row, want in golden.items():                                               # the reference's own vectors
        assert np.array_equal(out[row].cpu().numpy().view(np.uint32).astype(np.uint64), want), f"filetest{row}"
    rows = sorted(set(np.random.default_rng(3).integers(0, batch, 62).tolist() + [4, batch - 1]))
    ah, bh, oh = (t[rows].cpu().numpy().view(np.uint32) for t in (a, b, out))
    for i, r in enumerate(rows):                                                   # sampled rows, bit for bit
        assert np.array_equal(oh[i].astype(np.uint64), oracle.ntt_multiply(ah[i], bh[i], p)), r
    for x in (1, p - 1, 123456789 % p, 3141592):                                   # every row, evaluation identity
        assert torch.equal(_eval_rows(torch, a, n, x, p) * _eval_rows(torch, b, n, x, p) % p, _eval_rows(torch, out, n_out, x, p)), x
    # linearity in the first operand on all rows: (a + a') b = a b + a' b
    a2 = a.roll(1, dims=0)
    s = ((a.to(torch.int64) + a2.to(torch.int64)) % p).to(torch.int32)
    out_s = torch.empty_like(out); out_2 = torch.empty_like(out)
    gpu_lib.check(L.ntt_poly_multiply_batched_device(s.data_ptr(), b.data_ptr(), out_s.data_ptr(), n, p, batch, n, n, n_out, None, 0, st))
    gpu_lib.check(L.ntt_poly_multiply_batched_device(a2.data_ptr(), b.data_ptr(), out_2.data_ptr(), n, p, batch, n, n, n_out, None, 0, st))
    torch.cuda.synchronize()
    for lo in range(0, batch, 8192):
        want = (out[lo:lo + 8192].to(torch.int64) + out_2[lo:lo + 8192].to(torch.int64)) % p
        assert torch.equal(out_s[lo:lo + 8192].to(torch.int64), want)


def test_64_large_modulus_products_every_row_against_the_oracle(gpu_lib, oracle, route):
    import torch
    L = gpu_lib.lib
    n, batch, p = 131072, 64, BIGP
    g = torch.Generator(device="cuda"); g.manual_seed(0xB16B00B5)
    a = torch.randint(0, p, (batch, n), generator=g, device="cuda", dtype=torch.int64)
    b = torch.randint(0, p, (batch, n), generator=g, device="cuda", dtype=torch.int64)
    out = torch.full((batch, 2 * n - 1), -1, dtype=torch.int64, device="cuda")
    gpu_lib.check(L.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, batch, n, n, 2 * n - 1,
                                                         None, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ah, bh, oh = (t.cpu().numpy().view(np.uint64) for t in (a, b, out))
    for r in range(batch):
        assert np.array_equal(oh[r], oracle.crt_ntt_multiply(ah[r], bh[r], p)), r


def test_one_product_of_2_pow_28_coefficients(gpu_lib):
    import torch
    L = gpu_lib.lib
    p, n = BIGP, 1 << 28
    n_out = 2 * n - 1
    st = torch.cuda.current_stream().cuda_stream
    ws_bytes = L.ntt64_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(n_out, dtype=torch.int64, device="cuda")

    def product(a, b):
        out.fill_(-1)
        gpu_lib.check(L.ntt64_poly_multiply_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, ws.data_ptr(), ws_bytes, st))
        torch.cuda.synchronize()
        assert int(out.min()) >= 0 and int(out.max()) < p

    g = torch.Generator(device="cuda"); g.manual_seed(28)
    a = torch.randint(0, p, (n,), generator=g, device="cuda", dtype=torch.int64)
    # 1. shift by a monomial: a * x^k
    k = 123456789
    e = torch.zeros(n, dtype=torch.int64, device="cuda"); e[k] = 1
    product(a, e)
    assert torch.equal(out[k:k + n], a) and not bool(out[:k].any()) and not bool(out[k + n:].any())
    # 2. all ones: coefficient i is min(i + 1, 2n - 1 - i)
    ones = torch.ones(n, dtype=torch.int64, device="cuda")
    product(ones, ones)
    idx = torch.arange(n_out, dtype=torch.int64, device="cuda")
    assert torch.equal(out, torch.minimum(idx + 1, n_out - idx))
    del idx, ones, e
    # 3. random operands: A(x) B(x) = C(x) at x = 1, x = -1 and two random points (a wrong coefficient survives a random
    #    point with probability below 2n / p = 2^-21)
    b = torch.randint(0, p, (n,), generator=g, device="cuda", dtype=torch.int64)
    product(a, b)
    rng = np.random.default_rng(2028)
    for x in (1, p - 1, int(rng.integers(2, p - 1)), int(rng.integers(2, p - 1))):
        assert evaluate(a, x, p) * evaluate(b, x, p) % p == evaluate(out, x, p), x
    # 4. the same product in its explicit four-step form -- the matrix passes the multi-rank run uses
    #    (ntt64_fourstep_rows_load / cols_product / rows_store; replaces main_mpi_t.cc:139-291), here with one rank holding
    #    the whole matrix -- must equal the direct product bit for bit
    from ntt_amd.fourstep import FourStep
    fs = FourStep(gpu_lib, p, n, torch.device("cuda", torch.cuda.current_device()))
    assert fs.world == 1 and fs.R * fs.C == 2 * n
    sa, sb = fs.column_slab_device(a), fs.column_slab_device(b)
    c = fs.multiply_slabs(sa, sb)
    torch.cuda.synchronize()
    assert torch.equal(c.reshape(-1)[:n_out], out)

