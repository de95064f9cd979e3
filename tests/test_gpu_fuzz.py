"""Seeded random sweep over sizes, moduli, batch counts and row strides through the device entry points,
each case against the reference algorithm (power-of-two n) or the 128-bit schoolbook product (other n)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# A longer soak of the same sweeps: NTT_FUZZ_SEED shifts every generator's seed, NTT_FUZZ_SCALE multiplies the case counts
# (the defaults are what the suite runs).
SEED_SHIFT = int(os.environ.get("NTT_FUZZ_SEED", "0"))
SCALE = max(1, int(os.environ.get("NTT_FUZZ_SCALE", "1")))


def seeded(seed):
    return np.random.default_rng(seed + SEED_SHIFT)

PRIMES32 = [7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769]
OTHER = [1337006139375617, 1000000007, 2**61 - 1, 2**64 - 59, 4294967311, 65537, 97, 2, 3, 6, 2**32]


def expected(oracle, a, b, p):
    n = a.size
    if n & (n - 1) == 0 and p in PRIMES32 and 2 * n <= (1 << 18):
        return oracle.ntt_multiply(a, b, p)
    return oracle.schoolbook_u64(a, b, p)


def test_random_shapes_u64_device(gpu_lib, oracle, route):
    import torch
    rng = seeded(20240607)
    L = gpu_lib.lib
    st = torch.cuda.current_stream().cuda_stream
    sizes = [1, 2, 3, 5, 31, 32, 33, 100, 255, 256, 257, 1000, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 6000, 8191, 8192]
    for case in range(60 * SCALE):
        n = int(rng.choice(sizes)) if case % 3 else int(rng.integers(1, 3000))
        p = int(rng.choice(PRIMES32)) if case % 2 else int(OTHER[case // 2 % len(OTHER)])
        batch = int(rng.integers(1, 4))
        sa, sb, so = n + int(rng.integers(0, 5)), n + int(rng.integers(0, 5)), 2 * n - 1 + int(rng.integers(0, 5))
        a = rng.integers(0, p, (batch, sa), dtype=np.uint64); b = rng.integers(0, p, (batch, sb), dtype=np.uint64)
        if case % 5 == 0:
            a[:, :n] = p - 1; b[:, :n] = p - 1
        da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
        out = torch.full((batch, so), -1, dtype=torch.int64, device="cuda")
        gpu_lib.check(L.ntt_poly_multiply_batched_device_u64(da.data_ptr(), db.data_ptr(), out.data_ptr(), n, p, batch, sa, sb, so,
                                                              None, 0, st))
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        for r in range(batch):
            want = expected(oracle, a[r, :n], b[r, :n], p)
            assert np.array_equal(res[r, :2 * n - 1], want), (case, n, p, batch, r)
            assert (res[r, 2 * n - 1:] == np.uint64(2**64 - 1)).all(), "wrote past 2n-1"


def test_random_shapes_u32_device(gpu_lib, oracle):
    import torch
    rng = seeded(77)
    L = gpu_lib.lib
    st = torch.cuda.current_stream().cuda_stream
    for case in range(40 * SCALE):
        n = int(rng.integers(1, 9000)) if case % 2 else int(2 ** rng.integers(0, 16))
        p = int(rng.choice(PRIMES32))
        batch = int(rng.integers(1, 5))
        sa, sb, so = n + int(rng.integers(0, 3)), n + int(rng.integers(0, 3)), 2 * n - 1 + int(rng.integers(0, 3))
        a = rng.integers(0, p, (batch, sa), dtype=np.uint32); b = rng.integers(0, p, (batch, sb), dtype=np.uint32)
        da = torch.from_numpy(a.view(np.int32)).cuda(); db = torch.from_numpy(b.view(np.int32)).cuda()
        out = torch.full((batch, so), -1, dtype=torch.int32, device="cuda")
        ws_bytes = L.ntt_workspace_bytes(n, p, batch)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
        gpu_lib.check(L.ntt_poly_multiply_batched_device(da.data_ptr(), db.data_ptr(), out.data_ptr(), n, p, batch, sa, sb, so,
                                                          ws.data_ptr(), ws_bytes, st))
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint32)
        for r in range(batch):
            want = expected(oracle, a[r, :n].astype(np.uint64), b[r, :n].astype(np.uint64), p)
            assert np.array_equal(res[r, :2 * n - 1].astype(np.uint64), want), (case, n, p, batch, r)
            assert (res[r, 2 * n - 1:] == np.uint32(2**32 - 1)).all(), "wrote past 2n-1"


def test_random_long_shapes_through_the_64bit_field(gpu_lib, oracle):
    """Lengths where the 64-bit route runs its tile passes and fused core (transform lengths 2^17 .. 2^19), ragged n,
    padded rows, a caller workspace of exactly the advertised size, values at the top of the range."""
    import torch
    big = 1337006139375617
    rng = seeded(646464)
    L = gpu_lib.lib
    st = torch.cuda.current_stream().cuda_stream
    for case in range(8 * SCALE):
        n = int(rng.integers(32769, 140000)) if case % 4 else [65536, 131072][case // 4 % 2]
        p = big if case % 2 == 0 else int(rng.choice([998244353, 469762049]))
        batch = int(rng.integers(1, 4))
        sa, sb, so = n + int(rng.integers(0, 7)), n + int(rng.integers(0, 7)), 2 * n - 1 + int(rng.integers(0, 7))
        a = rng.integers(0, p, (batch, sa), dtype=np.uint64); b = rng.integers(0, p, (batch, sb), dtype=np.uint64)
        if case == 2:
            a[:, :n] = p - 1; b[:, :n] = p - 1
        da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
        out = torch.full((batch, so), -1, dtype=torch.int64, device="cuda")
        ws_bytes = L.ntt64_workspace_bytes_batched(n, batch)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        gpu_lib.check(L.ntt64_poly_multiply_batched_device(da.data_ptr(), db.data_ptr(), out.data_ptr(), n, p, batch, sa, sb, so,
                                                           ws.data_ptr(), ws_bytes, st))
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        m = 1 << (n - 1).bit_length()
        for r in range(batch):
            if 2 * m <= (1 << 18):
                az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a[r, :n]; bz[:n] = b[r, :n]
                want = oracle.poly_multiply_u64(az, bz, p)[:2 * n - 1]
                assert np.array_equal(res[r, :2 * n - 1], want), (case, n, p, batch, r)
            else:   # beyond the oracle's transform lengths: evaluation identities at x = 1 and x = -1
                c = res[r, :2 * n - 1]
                def ev(v, sign):
                    return (int(v[0::2].astype(object).sum()) + sign * int(v[1::2].astype(object).sum())) % p
                assert int(c.max()) < p
                for sign in (1, -1):
                    assert ev(a[r, :n], sign) * ev(b[r, :n], sign) % p == ev(c, sign), (case, n, p, sign)
            assert (res[r, 2 * n - 1:] == np.uint64(2**64 - 1)).all(), "wrote past 2n-1"
        too_small = max(ws_bytes - 8, 0)
        assert L.ntt64_poly_multiply_batched_device(da.data_ptr(), db.data_ptr(), out.data_ptr(), n, p, batch, sa, sb, so,
                                                    ws.data_ptr(), too_small, st) == -4


def test_both_large_modulus_routes_agree_on_random_shapes(gpu_lib):
    """The direct 64-bit field and the four-prime CRT are independent computations of the same residues (different
    kernels, different arithmetic: exact doubles against 32-bit Montgomery + 128-bit merge): 150 random shapes up to
    n = 2^17, ragged lengths, padded rows, values up to p - 1, compared with each other bit for bit."""
    import torch
    big = 1337006139375617
    rng = seeded(31337)
    L = gpu_lib.lib
    st = torch.cuda.current_stream().cuda_stream
    try:
        for case in range(150 * SCALE):
            hi = [64, 5000, 131072][case % 3]
            n = int(rng.integers(1, hi + 1)) if case % 7 else int(2 ** rng.integers(0, hi.bit_length()))
            batch = int(rng.integers(1, 6)) if n > 20000 else int(rng.integers(1, 40))
            sa, sb, so = n + int(rng.integers(0, 4)), n + int(rng.integers(0, 4)), 2 * n - 1 + int(rng.integers(0, 4))
            a = torch.randint(0, big, (batch, sa), device="cuda", dtype=torch.int64)
            b = torch.randint(0, big, (batch, sb), device="cuda", dtype=torch.int64)
            if case % 10 == 0:
                a.fill_(big - 1); b.fill_(big - 1)
            outs = []
            for policy in ("auto", "reference"):
                gpu_lib.set_route_policy(policy)
                out = torch.full((batch, so), -1, dtype=torch.int64, device="cuda")
                gpu_lib.check(L.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, big, batch, sa, sb, so,
                                                                     None, 0, st))
                outs.append(out)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[1]), (case, n, batch)
            assert int(outs[0][:, :2 * n - 1].max()) < big and int(outs[0][:, :2 * n - 1].min()) >= 0
            assert bool((outs[0][:, 2 * n - 1:] == -1).all())
    finally:
        gpu_lib.set_route_policy("auto")


def test_random_natural_order_transforms(gpu_lib, oracle):
    """ntt_transform_device_u32 / _u64 on random lengths 2 .. 2^16, batch counts that fill neither a workgroup nor a
    wavefront, both directions: every row against the reference's NTT() (main_naive.cc:96-137) for the shorter calls, a
    sample of rows for the longer ones, and inverse(forward(x)) = x for all of them."""
    import torch
    rng = seeded(424242)
    L = gpu_lib.lib
    st = torch.cuda.current_stream().cuda_stream
    for case in range(40 * SCALE):
        lg = int(rng.integers(1, 17))
        n = 1 << lg
        p = int(rng.choice(PRIMES32))
        batch = int(rng.integers(1, max(2, min(300, (1 << 18) >> lg))))
        inv = int(rng.integers(0, 2))
        x = rng.integers(0, p, (batch, n), dtype=np.uint64)
        if case % 4 == 0:
            x[batch - 1, :] = p - 1
        rows = range(batch) if batch * n <= (1 << 14) else sorted({0, batch // 2, batch - 1})
        t32 = torch.from_numpy(x.astype(np.uint32).view(np.int32)).cuda()
        t64 = torch.from_numpy(x.view(np.int64).copy()).cuda()
        gpu_lib.check(L.ntt_transform_device_u32(t32.data_ptr(), n, batch, inv, p, st))
        gpu_lib.check(L.ntt_transform_device_u64(t64.data_ptr(), n, batch, inv, p, st))
        torch.cuda.synchronize()
        got32 = t32.cpu().numpy().view(np.uint32).astype(np.uint64)
        got64 = t64.cpu().numpy().view(np.uint64)
        assert np.array_equal(got32, got64), (case, lg, p, batch, inv)
        for r in rows:
            assert np.array_equal(got32[r], oracle.ntt(x[r], bool(inv), p)), (case, lg, p, batch, inv, r)
        gpu_lib.check(L.ntt_transform_device_u32(t32.data_ptr(), n, batch, 1 - inv, p, st))
        torch.cuda.synchronize()
        assert np.array_equal(t32.cpu().numpy().view(np.uint32).astype(np.uint64), x), (case, lg, p, batch, inv)
