"""The library's arithmetic primitives one at a time on the GPU, against Python integers, on edge vectors: Montgomery
reduction (Montgomery32::REDC, Montgomery32.h:43-52, here without the last subtraction), the residue step that replaces
`a[j] % MOD` / Barrett::reduce (main_naive.cc:206-209, barrett.cc:22-30), the lazy butterflies, the 64-bit-field multiply
and reduction, and the reciprocal-based 128-by-64 remainder of the CRT merge (main_naive.cc:216-223).  A test-only kernel
library (tests/gpu_prims) applies each device function to every element of an input vector.  The Garner recombination is
driven through the public entry point with products of constants (n = 1) modulo 2^64 - 59, whose residues hit every borrow
path; 66 000 such rows also exercise the row loop of the per-coefficient kernels beyond the 65535-row launch limit."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
(OP_REDC_LAZY, OP_MULW_LZ, OP_MULW_NLZ, OP_REDUCE64_LZ, OP_REDUCE64_NLZ, OP_DIF_LZ, OP_DIF_NLZ, OP_DIT_LZ, OP_DIT_NLZ, OP_F64_MULW,
 OP_F64_REDUCE, OP_F64_DIF, OP_F64_DIT, OP_REM_2BY1, OP_REM128, OP_CANON_INV_LZ, OP_SUB_LZ, OP_FD_MULW, OP_FD_REDUCE, OP_FD_FROM_RAW,
 OP_FD_MEM, OP_FD_FROM_MEM, OP_FD_DIF, OP_FD_DIT) = range(24)


class Prims:
    def __init__(self):
        path = os.path.join(ROOT, "tests", "gpu_prims", "libnttprims.so")
        assert os.path.exists(path), "tests/gpu_prims/libnttprims.so missing: run __graft_entry__.build()"
        self.lib = ctypes.CDLL(path)
        self.lib.prims_consts_size.restype = ctypes.c_size_t
        vp = ctypes.c_void_p
        self.lib.prims_run.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, vp]
        self.lib.prims_run.restype = ctypes.c_int
        u32, u64 = ctypes.c_uint32, ctypes.c_uint64
        self.lib.prims_consts_fill.argtypes = [vp, u32, u32, u32, u32, u64, u64, u64, u64, u64, u64, ctypes.c_int]
        self.lib.prims_consts_fill.restype = None

    def consts(self, p32=998244353, p64=1337006139375617, pdiv=1337006139375617):
        buf = ctypes.create_string_buffer(self.lib.prims_consts_size())
        R32, R64 = 1 << 32, 1 << 64
        shift = 64 - pdiv.bit_length()
        d = pdiv << shift
        v = ((1 << 128) - 1) // d - (1 << 64)
        self.lib.prims_consts_fill(buf, p32, (-pow(p32, -1, R32)) % R32, R32 % p32, R32 * R32 % p32,
                                   p64, (-pow(p64, -1, R64)) % R64, R64 % p64, R64 * R64 % p64, d, v, shift)
        return buf

    def run(self, op, consts, *ins, two=False):
        import torch
        ts = [torch.from_numpy(np.asarray(x, dtype=np.uint64).view(np.int64)).cuda() for x in ins]
        n = ts[0].numel()
        out = torch.zeros(n, dtype=torch.int64, device="cuda"); out2 = torch.zeros(n, dtype=torch.int64, device="cuda")
        ptr = [t.data_ptr() for t in ts] + [None] * (3 - len(ts))
        assert self.lib.prims_run(op, ptr[0], ptr[1], ptr[2], out.data_ptr(), out2.data_ptr() if two else None, n, consts, None) == 0
        torch.cuda.synchronize()
        o = [int(x) for x in out.cpu().numpy().view(np.uint64)]
        return (o, [int(x) for x in out2.cpu().numpy().view(np.uint64)]) if two else o


@pytest.fixture(scope="module")
def prims(gpu_lib):
    return Prims()


def _edge32(p, rng, extra=()):
    vals = [0, 1, 2, p - 2, p - 1, p, p + 1, 2 * p - 1, 2 * p, 2 * p + 1, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, (1 << 32) - 2,
            3 * p if 3 * p < 1 << 32 else 5, 4 * p - 1 if 4 * p - 1 < 1 << 32 else 7] + list(extra)
    return [v for v in vals if 0 <= v < 1 << 32] + [int(x) for x in rng.integers(0, 1 << 32, 200, dtype=np.uint64)]


@pytest.mark.parametrize("p", [7340033, 469762049, 998244353, 1004535809, 1224736769])
def test_montgomery32_reduction_and_twiddle_multiply(prims, p):
    rng = np.random.default_rng(p)
    c = prims.consts(p32=p)
    R = 1 << 32
    Rinv = pow(R, -1, p)
    # redc_lazy: any T < 2^32 * p lands in [0, 2p) and is T * R^-1 mod p
    T = [0, 1, p, R - 1, R, R + 1, p * R - 1, (p - 1) * R, (R - 1) * (p - 1), (2 * p - 1) * (p - 1)] + \
        [int(a) * int(b) for a, b in zip(rng.integers(0, R, 300, dtype=np.uint64), rng.integers(0, p, 300, dtype=np.uint64))]
    T = [t for t in T if t < p * R]
    got = prims.run(OP_REDC_LAZY, c, T)
    assert all(g < 2 * p and g % p == t * Rinv % p for g, t in zip(got, T))
    lazy = p < 1 << 30
    xs = _edge32(p, rng)
    ws = [0, 1, p - 1, p // 2] + [int(x) for x in rng.integers(0, p, len(xs) - 4, dtype=np.uint64)]
    got = prims.run(OP_MULW_LZ if lazy else OP_MULW_NLZ, c, xs, ws)
    bound = 2 * p if lazy else p
    assert all(g < bound and g % p == x * w * Rinv % p for g, x, w in zip(got, xs, ws))


@pytest.mark.parametrize("p", [7340033, 998244353, 1004535809, 1224736769])
def test_residue_of_64_bit_words(prims, p):
    """reduce64 == x % p (the reference's `a[j] % MOD` / Barrett::reduce) up to the lazy range, for any 64-bit x."""
    rng = np.random.default_rng(p + 1)
    c = prims.consts(p32=p)
    xs = [0, 1, p - 1, p, p + 1, 2 * p, 3 * p + 5, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63) - 1, 1 << 63, (1 << 64) - 1,
          (1 << 64) - 2, p << 32, (p << 32) - 1, ((1 << 32) - 1) << 32, 1337006139375616] + \
         [int(x) for x in rng.integers(0, 1 << 64, 500, dtype=np.uint64)]
    lazy = p < 1 << 30
    got = prims.run(OP_REDUCE64_LZ if lazy else OP_REDUCE64_NLZ, c, xs)
    bound = 2 * p if lazy else p
    assert all(g < bound and g % p == x % p for g, x in zip(got, xs))


@pytest.mark.parametrize("p", [469762049, 998244353, 1224736769])
def test_lazy_butterflies_keep_their_ranges(prims, p):
    rng = np.random.default_rng(p + 2)
    c = prims.consts(p32=p)
    lazy = p < 1 << 30
    Rinv = pow(1 << 32, -1, p)
    B = 2 * p if lazy else p
    edge = [0, 1, p - 1, p, B - 1, B - 2]
    edge = [e for e in edge if e < B]
    us = edge * len(edge) + [int(x) for x in rng.integers(0, B, 300, dtype=np.uint64)]
    vs = [e for e in edge for _ in edge] + [int(x) for x in rng.integers(0, B, 300, dtype=np.uint64)]
    ws = [int(x) for x in rng.integers(1, p, len(us), dtype=np.uint64)]
    got = prims.run(OP_DIF_LZ if lazy else OP_DIF_NLZ, c, [u | v << 32 for u, v in zip(us, vs)], ws)
    for g, u, v, w in zip(got, us, vs, ws):
        u2, v2 = g & 0xFFFFFFFF, g >> 32
        assert u2 < B and v2 < B and u2 % p == (u + v) % p and v2 % p == (u - v) * w * Rinv % p
    # inverse butterfly: inputs in [0, 4p) (lazy) / [0, p)
    Bi = 4 * p if lazy else p
    edge = [e for e in [0, 1, p, 2 * p - 1, 2 * p, 3 * p, Bi - 1] if e < Bi]
    xs = edge * len(edge) + [int(x) for x in rng.integers(0, Bi, 300, dtype=np.uint64)]
    ys = [e for e in edge for _ in edge] + [int(x) for x in rng.integers(0, Bi, 300, dtype=np.uint64)]
    ws = [int(x) for x in rng.integers(1, p, len(xs), dtype=np.uint64)]
    got = prims.run(OP_DIT_LZ if lazy else OP_DIT_NLZ, c, [x | y << 32 for x, y in zip(xs, ys)], ws)
    for g, x, y, w in zip(got, xs, ys, ws):
        x2, y2 = g & 0xFFFFFFFF, g >> 32
        t = y * w * Rinv
        assert x2 < Bi and y2 < Bi and x2 % p == (x + t) % p and y2 % p == (x - t) % p
    if lazy:
        xs = edge + [int(x) for x in rng.integers(0, Bi, 300, dtype=np.uint64)]
        assert prims.run(OP_CANON_INV_LZ, c, xs) == [x % p for x in xs]
        us = [int(x) for x in rng.integers(0, B, 300, dtype=np.uint64)] + [0, B - 1, 0, 5]
        vs = [int(x) for x in rng.integers(0, B, 300, dtype=np.uint64)] + [B - 1, 0, 0, 5]
        got = prims.run(OP_SUB_LZ, c, [u | v << 32 for u, v in zip(us, vs)])
        assert all(g < B and g % p == (u - v) % p for g, u, v in zip(got, us, vs))


@pytest.mark.parametrize("p", [1337006139375617, 576460451655712769, 998244353, (1 << 59) + 0x1A5 * (1 << 32) + 1])
def test_64_bit_field_multiply_and_reduction(prims, p):
    if p % 2 == 0:
        pytest.skip("odd moduli only")
    rng = np.random.default_rng(p % 1000)
    c = prims.consts(p64=p)
    R = 1 << 64
    Rinv = pow(R, -1, p)
    xs = [0, 1, p - 1, p, 2 * p - 1, 2 * p, 4 * p - 1, R - 1, R - 2, 1 << 63] + [int(x) for x in rng.integers(0, 1 << 64, 400, dtype=np.uint64)]
    xs = [x for x in xs if x < R]
    ws = [0, 1, p - 1, 2 * p - 1] + [int(x) for x in rng.integers(0, 2 * p, len(xs) - 4, dtype=np.uint64)]
    pairs = [(x, w) for x, w in zip(xs, ws) if x * w < R * p]   # the precondition of mulw
    got = prims.run(OP_F64_MULW, c, [x for x, _ in pairs], [w for _, w in pairs])
    assert all(g < 2 * p and g % p == x * w * Rinv % p for g, (x, w) in zip(got, pairs))
    got = prims.run(OP_F64_REDUCE, c, xs)
    assert got == [x % p for x in xs]
    # butterflies: forward [0,2p) -> [0,2p), inverse [0,4p) -> [0,4p)
    us = [0, 2 * p - 1, 0, 2 * p - 1, p] + [int(x) for x in rng.integers(0, 2 * p, 300, dtype=np.uint64)]
    vs = [0, 2 * p - 1, 2 * p - 1, 0, p] + [int(x) for x in rng.integers(0, 2 * p, 300, dtype=np.uint64)]
    wm = [int(x) for x in rng.integers(0, p, len(us), dtype=np.uint64)]
    g1, g2 = prims.run(OP_F64_DIF, c, us, vs, wm, two=True)
    for a, b, u, v, w in zip(g1, g2, us, vs, wm):
        assert a < 2 * p and b < 2 * p and a % p == (u + v) % p and b % p == (u - v) * w * Rinv % p
    xs4 = [0, 4 * p - 1, 0, 4 * p - 1, 2 * p] + [int(x) for x in rng.integers(0, 4 * p, 300, dtype=np.uint64)]
    ys4 = [0, 4 * p - 1, 4 * p - 1, 0, 2 * p] + [int(x) for x in rng.integers(0, 4 * p, 300, dtype=np.uint64)]
    g1, g2 = prims.run(OP_F64_DIT, c, xs4, ys4, wm, two=True)
    for a, b, x, y, w in zip(g1, g2, xs4, ys4, wm):
        t = y * w * Rinv
        assert a < 4 * p and b < 4 * p and a % p == (x + t) % p and b % p == (x - t) % p


def _signed(v):
    return [x - (1 << 64) if x >= 1 << 63 else x for x in v]


@pytest.mark.parametrize("p", [1337006139375617, 998244353, 7340033, 1499999999999989])
def test_double_precision_field_is_exact_at_its_range_limits(prims, p):
    """FieldD (exact integer arithmetic in doubles, moduli up to 1.5e15): every primitive against Python integers at the
    edges of the ranges its butterflies rely on -- |x| up to 2.5 p into a multiply with |w| up to 0.6 p, anything below 2^52
    into a reduction -- plus the memory format (reduced value + p) and the canonical store."""
    rng = np.random.default_rng(p % 991)
    c = prims.consts(p64=p)
    half, big = p // 2, int(2.5 * p)
    def rnd(lo, hi, k):
        return [int(v) + lo for v in rng.integers(0, hi - lo, k, dtype=np.uint64)]
    xs = [0, 1, -1, big, -big, big - 1, p, -p, half, -half, p - 1, 1 - p] + rnd(-big, big, 2000)
    wmax = int(0.6 * p)
    ws = [wmax, -wmax, wmax, -wmax, half, -half, 1, -1, half, half, -half, half + 1] + rnd(-wmax, wmax, 2000)
    got = _signed(prims.run(OP_FD_MULW, c, [x % (1 << 64) for x in xs], [w % (1 << 64) for w in ws]))
    assert all(abs(g) <= 1.25 * p + 2 and (g - x * w) % p == 0 for g, x, w in zip(got, xs, ws))
    lim = (1 << 52) - 1
    xr = [0, lim, -lim, p, -p, half, half + 1, -half - 1, 3 * half] + rnd(-lim, lim, 2000)
    got = _signed(prims.run(OP_FD_REDUCE, c, [x % (1 << 64) for x in xr]))
    assert all(abs(g) <= half + 1 and (g - x) % p == 0 for g, x in zip(got, xr))
    raw = [0, 1, p - 1, p, p + 1, (1 << 40) - 1, 1 << 40, (1 << 52), (1 << 63), (1 << 64) - 1, (1 << 64) - 2] + \
          [int(v) for v in rng.integers(0, 1 << 64, 2000, dtype=np.uint64)]
    got = _signed(prims.run(OP_FD_FROM_RAW, c, raw))
    assert all(abs(g) <= half + 1 and (g - x) % p == 0 for g, x in zip(got, raw))
    # memory format and canonical store, for every value a kernel may hold (|x| < 2^52)
    m, cn = prims.run(OP_FD_MEM, c, [x % (1 << 64) for x in xr], two=True)
    assert all(half <= a <= 3 * half + 2 and (a - x) % p == 0 and b == x % p for a, b, x in zip(m, cn, xr))
    back = _signed(prims.run(OP_FD_FROM_MEM, c, m))
    assert all(abs(g) <= half + 1 and (g - x) % p == 0 for g, x in zip(back, xr))
    canon_words = [0, 1, p - 1, 2 * p - 1, p] + [int(v) for v in rng.integers(0, 2 * p, 500, dtype=np.uint64)]   # words other kernels store
    back = _signed(prims.run(OP_FD_FROM_MEM, c, canon_words))
    assert all(abs(g) <= p and (g - x) % p == 0 for g, x in zip(back, canon_words))
    # butterflies: forward inputs up to 1.25 p, inverse inputs up to 1.75 p / 2.5 p
    b125, b175 = int(1.25 * p), int(1.75 * p)
    us = [b125, -b125, b125, -b125] + rnd(-b125, b125, 1000); vs = [b125, -b125, -b125, b125] + rnd(-b125, b125, 1000)
    wt = [wmax, -wmax, wmax, -wmax] + rnd(-wmax, wmax, 1000)
    g1, g2 = prims.run(OP_FD_DIF, c, [x % (1 << 64) for x in us], [x % (1 << 64) for x in vs], [x % (1 << 64) for x in wt], two=True)
    for a, b, u, v, w in zip(_signed(g1), _signed(g2), us, vs, wt):
        assert abs(a) <= half + 1 and abs(b) <= b125 + 2 and (a - (u + v)) % p == 0 and (b - (u - v) * w) % p == 0
    xs2 = [lim, -lim, b175, -b175] + rnd(-b175, b175, 1000); ys2 = [big, -big, big, -big] + rnd(-big, big, 1000)
    g1, g2 = prims.run(OP_FD_DIT, c, [x % (1 << 64) for x in xs2], [x % (1 << 64) for x in ys2], [x % (1 << 64) for x in wt], two=True)
    for a, b, x, y, w in zip(_signed(g1), _signed(g2), xs2, ys2, wt):
        assert abs(a) <= b175 + 3 and abs(b) <= b175 + 3 and (a - (x + y * w)) % p == 0 and (b - (x - y * w)) % p == 0


@pytest.mark.parametrize("p", [2, 3, 1337006139375617, (1 << 63) - 25, 1 << 63, (1 << 64) - 59, (1 << 64) - 1, (1 << 32) + 15, 998244353])
def test_remainder_of_128_by_64_bits(prims, p):
    """rem_2by1 / rem128: (u1:u0) mod p for u1 < p, including shift = 0 (top bit of p set) and p near 2^64."""
    rng = np.random.default_rng(p % 977)
    c = prims.consts(pdiv=p)
    u1 = [0, p - 1, 0, p - 1, p // 2] + [int(x) % p for x in rng.integers(0, 1 << 64, 500, dtype=np.uint64)]
    u0 = [0, (1 << 64) - 1, (1 << 64) - 1, 0, p] + [int(x) for x in rng.integers(0, 1 << 64, 500, dtype=np.uint64)]
    u0 = [x % (1 << 64) for x in u0]
    want = [((a << 64) | b) % p for a, b in zip(u1, u0)]
    assert prims.run(OP_REM_2BY1, c, u1, u0) == want
    assert prims.run(OP_REM128, c, u1, u0) == want


def test_garner_borrow_paths_and_row_loop_through_the_public_entry_point(gpu_lib):
    """Products of constants modulo 2^64 - 59 go through three 59-bit primes and the Garner recombination; the residues of
    x * y modulo the three primes are unrelated, so r2 < t1, r3 < t1 and u < t2 all occur.  66 000 rows > 65535."""
    import torch
    p = (1 << 64) - 59
    rng = np.random.default_rng(59)
    rows = 66000
    x = rng.integers(0, 1 << 64, rows, dtype=np.uint64); y = rng.integers(0, 1 << 64, rows, dtype=np.uint64)
    x[:6] = [0, 1, p - 1, p - 1, 1 << 63, 576460451655712769]; y[:6] = [5, 1, p - 1, 1, 2, 576460065108656129]
    x %= np.uint64(p); y %= np.uint64(p)
    a = torch.from_numpy(x.view(np.int64)).cuda(); b = torch.from_numpy(y.view(np.int64)).cuda()
    out = torch.zeros(rows, dtype=torch.int64, device="cuda")
    gpu_lib.check(gpu_lib.lib.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), 1, p, rows, 1, 1, 1, None, 0, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    want = np.array([int(u) * int(v) % p for u, v in zip(x, y)], dtype=np.uint64)
    assert np.array_equal(got, want)
    # the four-prime merge kernel with more rows than one grid dimension takes: 70 000 products of two coefficients
    P, rows = 1337006139375617, 70000
    A = rng.integers(0, P, (rows, 2), dtype=np.uint64); Bm = rng.integers(0, P, (rows, 2), dtype=np.uint64)
    ta = torch.from_numpy(A.view(np.int64)).cuda(); tb = torch.from_numpy(Bm.view(np.int64)).cuda()
    o = torch.zeros((rows, 3), dtype=torch.int64, device="cuda")
    gpu_lib.check(gpu_lib.lib.ntt_poly_multiply_batched_device_u64(ta.data_ptr(), tb.data_ptr(), o.data_ptr(), 2, P, rows, 2, 2, 3, None, 0, None))
    torch.cuda.synchronize()
    Ao, Bo = A.astype(object), Bm.astype(object)
    want = np.stack([Ao[:, 0] * Bo[:, 0] % P, (Ao[:, 0] * Bo[:, 1] + Ao[:, 1] * Bo[:, 0]) % P, Ao[:, 1] * Bo[:, 1] % P], axis=1).astype(np.uint64)
    assert np.array_equal(o.cpu().numpy().view(np.uint64), want)
