"""GPU parity of the 64-bit-field path (direct product and explicit four-step), through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BIGP = 1337006139375617  # 19 * 2^46 + 1, the reference's large modulus (main_naive.cc:234)


def gpu_mul64(lib, a, b, p):
    import torch
    n = len(a)
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()
    tb = torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint64).view(np.int64)).cuda()
    out = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
    lib.check(lib.lib.ntt64_poly_multiply_device(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, p, None, 0,
                                                 torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("p", [BIGP, 998244353, 469762049])
@pytest.mark.parametrize("n", [1, 2, 3, 8, 100, 2048, 4096, 5000, 8192, 65536, 131072])
def test_direct_64bit_product_matches_oracle(gpu_lib, oracle, p, n):
    rng = np.random.default_rng(n + p % 13)
    a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
    if n & (n - 1) == 0:
        want = oracle.poly_multiply_u64(a, b, p)   # CRT route of the reference for BIGP, plain NTT otherwise
    else:
        m = 1 << (n - 1).bit_length()
        az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a; bz[:n] = b
        want = oracle.poly_multiply_u64(az, bz, p)[:2 * n - 1]
    assert np.array_equal(gpu_mul64(gpu_lib, a, b, p), want)


def test_64bit_and_crt_routes_agree_on_golden_vectors(gpu_lib):
    from golden_util import iter_golden
    for name, n, p, a, b, out in iter_golden():
        if name.startswith("ref_crt") or name.startswith("nttdata1"):
            assert np.array_equal(gpu_mul64(gpu_lib, a, b, p), out), name


def test_extreme_and_unreduced_inputs(gpu_lib, oracle):
    a = np.full(131072, BIGP - 1, dtype=np.uint64)
    assert np.array_equal(gpu_mul64(gpu_lib, a, a, BIGP), oracle.crt_ntt_multiply(a, a, BIGP))
    rng = np.random.default_rng(9)
    a = rng.integers(0, 2**64, 300, dtype=np.uint64); b = rng.integers(0, 2**64, 300, dtype=np.uint64)
    assert np.array_equal(gpu_mul64(gpu_lib, a, b, BIGP), oracle.schoolbook_u64(a % BIGP, b % BIGP, BIGP))


@pytest.mark.parametrize("p,n,batch", [(BIGP, 1, 3), (BIGP, 100, 7), (998244353, 2048, 33), (BIGP, 4096, 16), (BIGP, 5000, 3),
                                       (BIGP, 65536, 5), (998244353, 40000, 2), (BIGP, 131072, 9), (BIGP, 131072, 130)])
def test_batched_products_in_the_field(gpu_lib, oracle, p, n, batch):
    import torch
    from test_cpu_twin import check_batched_field_product
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda()
    from_dev = lambda t: t.cpu().numpy().view(np.uint64).copy()
    if batch > 100:   # more rows than one group of the workspace: every row against a few distinct operands
        rng = np.random.default_rng(5)
        a0 = rng.integers(0, p, (3, n), dtype=np.uint64); b0 = rng.integers(0, p, (3, n), dtype=np.uint64)
        want = [oracle.poly_multiply_u64(a0[i], b0[i], p) for i in range(3)]
        idx = np.arange(batch) % 3
        out = torch.zeros((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
        ta, tb = to_dev(a0[idx]), to_dev(b0[idx])
        gpu_lib.check(gpu_lib.lib.ntt64_poly_multiply_batched_device(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, p, batch, n, n,
                                                                     2 * n - 1, None, 0, torch.cuda.current_stream().cuda_stream))
        got = from_dev(out).reshape(batch, 2 * n - 1)
        for r in range(batch):
            assert np.array_equal(got[r], want[idx[r]]), r
        return
    check_batched_field_product(gpu_lib, oracle, to_dev, from_dev, torch.cuda.current_stream().cuda_stream, p, n, batch)


@pytest.mark.parametrize("n", [7, 4096, 30000, 131072])
def test_explicit_fourstep_single_gpu(gpu_lib, oracle, n):
    import torch
    from ntt_amd.fourstep import poly_multiply_fourstep
    rng = np.random.default_rng(n)
    a = rng.integers(0, BIGP, n, dtype=np.uint64); b = rng.integers(0, BIGP, n, dtype=np.uint64)
    m = 1 << (n - 1).bit_length()
    az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a; bz[:n] = b
    want = oracle.crt_ntt_multiply(az, bz, BIGP)[:2 * n - 1]
    assert np.array_equal(poly_multiply_fourstep(gpu_lib, a, b, BIGP, torch.device("cuda", 0)), want)


@pytest.mark.parametrize("world,n", [(2, 5), (2, 3000), (4, 20000), (8, 70000), (16, 9000), (8, 131072), (2, 100000),
                                     (2, 2), (4, 8), (8, 32), (16, 128)])   # the last four: slabs one column wide
def test_fourstep_product_with_emulated_ranks(gpu_lib, oracle, world, n):
    """The multi-rank four-step product with every rank emulated on this one GPU (exchanges as tensor copies): the pass over
    the parts and the transforms on sub-ranges of a part run as GPU kernels here, not only on real multi-GPU nodes."""
    import torch
    from fourstep_emulation import emulated_fourstep_product
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda()
    from_dev = lambda t: t.cpu().numpy().view(np.uint64).copy()
    rng = np.random.default_rng(world * 1000 + n)
    a = rng.integers(0, 2**64, n, dtype=np.uint64); b = rng.integers(0, BIGP, n, dtype=np.uint64)
    m = 1 << (n - 1).bit_length()
    az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a % np.uint64(BIGP); bz[:n] = b
    want = oracle.crt_ntt_multiply(az, bz, BIGP)[:2 * n - 1]
    got = emulated_fourstep_product(gpu_lib, a, b, BIGP, world, to_dev, from_dev, torch, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(got, want)


def test_emulated_eight_rank_product_of_2_pow_24_equals_the_direct_one(gpu_lib):
    """n = 2^24 over eight emulated ranks (matrix 2^12 x 2^13: tile passes over the rows, parts pass + fused core on the
    columns) against the single-GPU direct product."""
    import torch
    from fourstep_emulation import emulated_fourstep_product
    n = 1 << 24
    rng = np.random.default_rng(24)
    a = rng.integers(0, BIGP, n, dtype=np.uint64); b = rng.integers(0, BIGP, n, dtype=np.uint64)
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda()
    from_dev = lambda t: t.cpu().numpy().view(np.uint64).copy()
    got_all = emulated_fourstep_product(gpu_lib, a, b, BIGP, 8, to_dev, from_dev, torch, torch.cuda.current_stream().cuda_stream, pieces=4, all_rows=True)
    got = emulated_fourstep_product(gpu_lib, a, b, BIGP, 8, to_dev, from_dev, torch, torch.cuda.current_stream().cuda_stream, pieces=4)
    assert np.array_equal(got, got_all)      # the loading pass with and without its shortcut for the padding rows
    assert np.array_equal(got, gpu_mul64(gpu_lib, a, b, BIGP))


def test_long_product_beyond_the_crt_bound(gpu_lib):
    """n = 2^20 with the 51-bit modulus: past the four-prime CRT bound, so only the 64-bit field applies.
    Checked exactly on structured inputs (shift by a monomial, all-ones) and at x = 1 and x = -1 on random ones."""
    p, n = BIGP, 1 << 20
    rng = np.random.default_rng(20)
    a = rng.integers(0, p, n, dtype=np.uint64)
    e = np.zeros(n, dtype=np.uint64); e[12345] = 1
    got = gpu_mul64(gpu_lib, a, e, p)
    assert np.array_equal(got[12345:12345 + n], a) and not got[:12345].any() and not got[12345 + n:].any()
    ones = np.ones(n, dtype=np.uint64)
    got = gpu_mul64(gpu_lib, ones, ones, p)
    k = np.arange(2 * n - 1, dtype=np.int64)
    assert np.array_equal(got.astype(np.int64), np.minimum(k + 1, 2 * n - 1 - k))
    b = rng.integers(0, p, n, dtype=np.uint64)
    c = gpu_mul64(gpu_lib, a, b, p)
    assert c.max() < p
    def at(v, sign):
        s = [int(x) for x in v.astype(object)] if False else None
        ev, od = int(v[0::2].astype(object).sum()), int(v[1::2].astype(object).sum())
        return (ev + sign * od) % p
    for sign in (1, -1):
        assert at(a, sign) * at(b, sign) % p == at(c, sign)


@pytest.mark.parametrize("p", [7340033, 998244353, 1224736769])
def test_standalone_transform_matches_reference_ntt(gpu_lib, oracle, p):
    # these moduli take the 32-bit kernels; the 64-bit-field transform is checked on the same moduli in the next test (the
    # reference's NTT(), which the oracle restates, is only defined for moduli below 2^32)
    rng = np.random.default_rng(p % 31)
    for n in (1, 2, 8, 512, 4096, 8192, 1 << 16, 1 << 18):
        x = rng.integers(0, p, n, dtype=np.uint64)
        fwd = gpu_lib.transform_u64(x, False, p)
        assert np.array_equal(fwd, oracle.ntt(x, False, p)), n
        assert np.array_equal(gpu_lib.transform_u64(x, True, p), oracle.ntt(x, True, p)), n
        assert np.array_equal(gpu_lib.transform_u64(fwd, True, p), x)


def test_standalone_transform_of_small_moduli_in_the_64_bit_field():
    """NTT_TRANSFORM32=off (fresh process: the switch is read once) keeps moduli below 2^31 on the 64-bit-field transform,
    which must give the same natural-order results."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from ntt_amd._binding import load_product_library
        from oracle_binding import Oracle
        lib = load_product_library(); lib.init(0); orc = Oracle()
        rng = np.random.default_rng(3)
        ok = True
        for p in (7340033, 998244353, 1224736769):
            for n in (2, 64, 4096, 8192, 1 << 16):
                x = rng.integers(0, p, n, dtype=np.uint64)
                ok = ok and np.array_equal(lib.transform_u64(x, False, p), orc.ntt(x, False, p))
                ok = ok and np.array_equal(lib.transform_u64(x, True, p), orc.ntt(x, True, p))
        print("RESULT", ok)
    """) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NTT_TRANSFORM32="off"), capture_output=True, text=True, timeout=300)
    assert "RESULT True" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


def test_standalone_transform_batched_device(gpu_lib, oracle):
    import torch
    p, n, batch = BIGP, 4096, 5
    rng = np.random.default_rng(1)
    x = rng.integers(0, p, (batch, n), dtype=np.uint64)
    t = torch.from_numpy(x.view(np.int64)).cuda()
    gpu_lib.check(gpu_lib.lib.ntt_transform_device_u64(t.data_ptr(), n, batch, 0, p, torch.cuda.current_stream().cuda_stream))
    gpu_lib.check(gpu_lib.lib.ntt_transform_device_u64(t.data_ptr(), n, batch, 1, p, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().view(np.uint64), x)


def test_transform_along_a_strided_axis_tiled_and_register_passes(gpu_lib, oracle):
    import torch
    from test_cpu_twin import check_middle_axis_transform
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda()
    from_dev = lambda t: t.cpu().numpy().view(np.uint64).copy()
    st = torch.cuda.current_stream().cuda_stream
    for log_len, inner, outer in [(10, 6, 2), (9, 8, 1), (11, 16, 1), (5, 3, 3), (13, 1, 2), (14, 1, 1), (4, 40, 2),
                                  (18, 1, 1), (9, 512, 1), (12, 24, 2)]:
        check_middle_axis_transform(gpu_lib, oracle, to_dev, from_dev, st, 23, log_len, inner, outer)


def test_rows_pass_of_a_one_column_slab_keeps_its_twiddles(gpu_lib):
    """Slabs one column wide (cols_local == 1, col0 != 0): the row passes must be ones that carry the inter-axis twiddles."""
    import torch
    from test_cpu_twin import check_one_column_slab_rows_pass
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda()
    from_dev = lambda t: t.cpu().numpy().view(np.uint64).copy()
    check_one_column_slab_rows_pass(gpu_lib, to_dev, from_dev, torch.cuda.current_stream().cuda_stream)
