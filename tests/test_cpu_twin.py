"""Host logic and kernel index arithmetic without a GPU: the library sources compiled by g++ over a
HIP stand-in (tests/cpu_twin) must reproduce the oracle bit for bit.  This exercises plan building,
dispatch by size and modulus, twiddle tables, the LDS slot mapping, the lazy-range bounds and the
128-bit CRT merge -- everything except the GPU code generation itself (covered by the -m gpu tests)."""
import os

import numpy as np
import pytest

from golden_util import iter_golden

BIGP = 1337006139375617


@pytest.mark.parametrize("p", [7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769])
def test_direct_path_all_size_classes(twin, oracle, p):
    rng = np.random.default_rng(p % 97)
    for n in (1, 2, 4, 8, 16, 32, 64, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
        a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
        assert np.array_equal(twin.multiply_u64(a, b, p), oracle.ntt_multiply(a, b, p)), n


@pytest.mark.parametrize("p", [7340033, 998244353, 1224736769])
def test_short_products_exchanged_inside_the_quad(twin, oracle, p):
    """transform lengths 32 and 64 (n = 9 .. 32): the levels that run on registers exchanged between neighbouring lanes,
    several products per workgroup, batches that do not fill the last workgroup"""
    rng = np.random.default_rng(p % 89)
    for n in (9, 12, 16, 17, 24, 31, 32):
        for batch in (1, 3, 130):
            a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
            a[0] = p - 1; b[0] = p - 1
            got = twin.multiply_batched(a, b, p)
            want = np.stack([oracle.schoolbook_u64(a[r].astype(np.uint64), b[r].astype(np.uint64), p) for r in range(batch)])
            assert np.array_equal(got.astype(np.uint64), want), (n, batch)


def test_golden_vectors(twin, route):
    for name, n, p, a, b, out in iter_golden():
        assert np.array_equal(twin.multiply_u64(a, b, p), out), name


def test_extreme_values_exercise_lazy_bounds(twin, oracle):
    for p in (998244353, 1004535809, 1224736769, 7340033):
        for n in (4096, 8192, 131072):
            a = np.full(n, p - 1, dtype=np.uint64)
            assert np.array_equal(twin.multiply_u64(a, a, p), oracle.ntt_multiply(a, a, p)), (p, n)


def test_crt_path(twin, oracle, route):
    rng = np.random.default_rng(4)
    for n in (1, 4, 1000, 4096, 8192, 131072):
        a = rng.integers(0, BIGP, n, dtype=np.uint64); b = rng.integers(0, BIGP, n, dtype=np.uint64)
        want = oracle.crt_ntt_multiply(a, b, BIGP) if n & (n - 1) == 0 else oracle.schoolbook_u64(a, b, BIGP)
        assert np.array_equal(twin.multiply_u64(a, b, BIGP), want), n
    a = np.full(131072, BIGP - 1, dtype=np.uint64)
    assert np.array_equal(twin.multiply_u64(a, a, BIGP), oracle.crt_ntt_multiply(a, a, BIGP))


def test_crt_merge_against_python_integers(twin):
    """the recombination kernel alone: Garner digits and both tails (exact doubles up to 1.5e15, 128-bit integers above)"""
    import crt_merge_cases as cm
    rng = np.random.default_rng(216)
    xs = cm.cases(rng, 1500)
    to_dev = lambda arr: {"arr": (a := np.ascontiguousarray(arr)), "ptr": a.ctypes.data}
    from_dev = lambda d: d["arr"]
    for p in cm.MODULI:
        cm.check(twin, to_dev, from_dev, p, xs)


@pytest.mark.parametrize("p,n", [(1000000007, 100), (97, 50), (2**31 - 1, 64), (4294967291, 16), (2**40 + 15, 300),
                                 (2**58 - 27, 4), (2, 10), (2**59 - 55, 1)])
def test_arbitrary_moduli_through_crt(twin, oracle, p, n):
    rng = np.random.default_rng(6)
    a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
    assert np.array_equal(twin.multiply_u64(a, b, p), oracle.schoolbook_u64(a, b, p))


def test_non_power_of_two_lengths(twin, oracle):
    p = 998244353
    rng = np.random.default_rng(7)
    for n in (3, 5, 100, 1000, 3000, 4097, 5000, 9000, 100000):
        a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
        m = 1 << (n - 1).bit_length()
        az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a; bz[:n] = b
        assert np.array_equal(twin.multiply_u64(a, b, p), oracle.ntt_multiply(az, bz, p)[:2 * n - 1]), n


def test_batched_and_i32_entry_points(twin, oracle, route):
    p = 998244353
    rng = np.random.default_rng(8)
    for n, batch in ((4096, 5), (1024, 9), (16384, 3)):
        a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
        got = twin.multiply_batched(a, b, p)
        for r in range(batch):
            assert np.array_equal(got[r].astype(np.uint64), oracle.ntt_multiply(a[r], b[r], p))
    a = rng.integers(0, BIGP, (3, 2048), dtype=np.uint64); b = rng.integers(0, BIGP, (3, 2048), dtype=np.uint64)
    got = twin.multiply_batched_u64(a, b, BIGP)
    for r in range(3):
        assert np.array_equal(got[r], oracle.crt_ntt_multiply(a[r], b[r], BIGP))
    assert twin.multiply_i32(np.array([4, 1, 5, 2]), np.array([1, 5, 5, 4]), 7340033).tolist() == [4, 21, 30, 48, 39, 30, 8]
    a = rng.integers(0, 1000000007, 40, dtype=np.int32); b = rng.integers(0, 1000000007, 40, dtype=np.int32)
    assert np.array_equal(twin.multiply_i32(a, b, 1000000007), oracle.schoolbook_i32(a, b, 1000000007))


def test_error_reporting(twin):
    from ntt_amd._binding import NttError
    one = np.ones(4, dtype=np.uint64)
    with pytest.raises(NttError) as e:
        twin.multiply_u64(one, one, 1)            # no modulus below 2
    assert e.value.status == -1
    with pytest.raises(NttError):
        twin.multiply_batched(np.ones((2, 8), dtype=np.uint32), np.ones((2, 8), dtype=np.uint32), 1000000007)
    assert twin.describe_plan(131072, 998244353) == "k_fwd_strided,k_core_mul,k_inv_strided"
    assert twin.describe_plan(4096, 998244353) == "k_fused8192"
    twin.set_route_policy("reference")
    assert twin.describe_plan(4, BIGP).endswith("k_crt_merge")
    assert twin.lib.ntt_workspace_bytes(131072, BIGP, 3) == 8 * 3 * (1 << 18) * 4    # both operands x four primes
    twin.set_route_policy("auto")
    assert twin.describe_plan(4, BIGP).startswith("k64_")
    assert twin.lib.ntt_workspace_bytes(1024, BIGP, 3) == twin.lib.ntt64_workspace_bytes_batched(1024, 3) == 2 * 3 * 2048 * 8
    assert twin.lib.ntt_workspace_bytes(131072, 998244353, 4) == 2 * 4 * (1 << 18) * 4


# ---- 64-bit field path -------------------------------------------------------------------------
def _twin_mul64(twin, a, b, p):
    import torch
    n = len(a)
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy())
    tb = torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint64).view(np.int64).copy())
    out = torch.zeros(2 * n - 1, dtype=torch.int64)
    twin.check(twin.lib.ntt64_poly_multiply_device(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, p, None, 0, None))
    return out.numpy().view(np.uint64)


def test_64bit_field_product(twin, oracle):
    rng = np.random.default_rng(64)
    for p in (BIGP, 998244353):
        for n in (1, 2, 3, 8, 100, 2048, 4096, 5000, 8192, 32768):
            a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
            m = 1 << (n - 1).bit_length() if n > 1 else 1
            az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a; bz[:n] = b
            want = oracle.poly_multiply_u64(az, bz, p)[:2 * n - 1]
            assert np.array_equal(_twin_mul64(twin, a, b, p), want), (p, n)
    a = rng.integers(0, 2**64, 300, dtype=np.uint64); b = rng.integers(0, 2**64, 300, dtype=np.uint64)
    assert np.array_equal(_twin_mul64(twin, a, b, BIGP), oracle.schoolbook_u64(a % BIGP, b % BIGP, BIGP))


def check_batched_field_product(lib, oracle, to_dev, from_dev, stream, p, n, batch, pad=(3, 0, 5)):
    """ntt64_poly_multiply_batched_device against the oracle: rows with strides longer than the rows, unreduced inputs."""
    rng = np.random.default_rng(n * 31 + batch)
    sa, sb, so = n + pad[0], n + pad[1], 2 * n - 1 + pad[2]
    a = rng.integers(0, 2**64, (batch, sa), dtype=np.uint64); b = rng.integers(0, p, (batch, sb), dtype=np.uint64)
    ta, tb = to_dev(a), to_dev(b)
    out = to_dev(np.full((batch, so), 7, dtype=np.uint64))
    ws = lib.lib.ntt64_workspace_bytes_batched(n, batch)
    assert ws >= 16 * (2 * n - 1)
    lib.check(lib.lib.ntt64_poly_multiply_batched_device(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, p, batch, sa, sb, so,
                                                         None, 0, stream))
    got = from_dev(out).reshape(batch, so)
    m = 1 << (n - 1).bit_length() if n > 1 else 1
    for r in range(batch):
        az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64)
        az[:n] = a[r, :n] % np.uint64(p); bz[:n] = b[r, :n]
        want = oracle.poly_multiply_u64(az, bz, p)[:2 * n - 1]
        assert np.array_equal(got[r, :2 * n - 1], want), (p, n, r)
        assert (got[r, 2 * n - 1:] == 7).all(), "padding between result rows was written"


def test_64bit_field_batched_products(twin, oracle):
    import torch
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy())
    from_dev = lambda t: t.numpy().view(np.uint64).copy()
    for p, n, batch in ((BIGP, 1, 3), (BIGP, 5, 4), (BIGP, 100, 3), (998244353, 2048, 5), (BIGP, 4096, 2), (BIGP, 5000, 3),
                        (BIGP, 65536, 3), (998244353, 40000, 2)):
        check_batched_field_product(twin, oracle, to_dev, from_dev, None, p, n, batch)


def test_64bit_field_rejects_unfriendly_moduli(twin):
    from ntt_amd._binding import NttError
    a = np.ones(8, dtype=np.uint64)
    with pytest.raises(NttError):
        _twin_mul64(twin, a, a, 1000000007)      # 16 does not divide p-1
    with pytest.raises(NttError):
        _twin_mul64(twin, a, a, 2**61 - 1)       # above the 2^60 limit
    # composite moduli with 16 | p-1 never reach the field kernels (their Fermat inverses would not invert): 1729 is a
    # Carmichael number, 4097 = 17 * 241, 2^32 + 1 = 641 * 6700417
    for composite in (1729, 4097, 2**32 + 1):
        with pytest.raises(NttError, match="not prime"):
            _twin_mul64(twin, a, a, composite)
    # ... while the general entry point still serves them, through a CRT route
    rng = np.random.default_rng(1729)
    from oracle_binding import Oracle
    for composite in (1729, 2**32 + 1):
        x = rng.integers(0, composite, 8, dtype=np.uint64); y = rng.integers(0, composite, 8, dtype=np.uint64)
        assert np.array_equal(twin.multiply_u64(x, y, composite), Oracle().schoolbook_u64(x, y, composite))


def test_explicit_fourstep_matrix_form(twin, oracle):
    import torch
    from ntt_amd.fourstep import poly_multiply_fourstep, plan_shape
    rng = np.random.default_rng(4)
    for n in (1, 2, 7, 64, 1000, 4096, 30000):
        a = rng.integers(0, BIGP, n, dtype=np.uint64); b = rng.integers(0, BIGP, n, dtype=np.uint64)
        m = 1 << (n - 1).bit_length() if n > 1 else 1
        az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a; bz[:n] = b
        want = oracle.crt_ntt_multiply(az, bz, BIGP)[:2 * n - 1]
        assert np.array_equal(poly_multiply_fourstep(twin, a, b, BIGP, torch.device("cpu")), want), n
    assert plan_shape(1 << 28, 8) == (29, 14, 15)
    assert plan_shape(1 << 28, 1) == (29, 17, 12)      # two tile passes over the rows, the column axis inside the fused core
    # the fused product (operand load in the first row pass, column passes + pointwise in one call, canonical store in the
    # last row pass) equals the product assembled from the separate transforms, on arbitrary 64-bit input words
    from ntt_amd.fourstep import FourStep
    for n in (3, 700, 9000, 40000):
        a = rng.integers(0, 2**64, n, dtype=np.uint64); b = rng.integers(0, 2**64, n, dtype=np.uint64)
        fs = FourStep(twin, BIGP, n, torch.device("cpu"))
        fused = fs.gather(fs.multiply_slabs(fs.column_slab(a), fs.column_slab(b)))
        stepwise = fs.gather(fs.multiply_slabs_stepwise(fs.column_slab(a), fs.column_slab(b)))
        assert np.array_equal(fused, stepwise), n
        assert np.array_equal(fused, oracle.schoolbook_u64(a % np.uint64(BIGP), b % np.uint64(BIGP), BIGP)) if n <= 9000 else True


def test_fourstep_product_with_emulated_ranks(twin, oracle):
    """All ranks of the sharded product in one process: the pass over the parts and the transforms on sub-ranges of a part."""
    import torch
    from fourstep_emulation import emulated_fourstep_product
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy())
    from_dev = lambda t: t.numpy().view(np.uint64).copy()
    rng = np.random.default_rng(44)
    # (2, 2), (4, 5), (4, 8), (8, 20), (8, 32), (16, 128): slabs one column wide (transform length = world^2), whose row passes
    # must still apply the twiddles of their column
    for world, n in ((2, 5), (2, 3000), (4, 20000), (8, 70000), (16, 9000), (1, 5000), (2, 2), (4, 5), (4, 8), (8, 20), (8, 32), (16, 128)):
        a = rng.integers(0, 2**64, n, dtype=np.uint64); b = rng.integers(0, BIGP, n, dtype=np.uint64)
        m = 1 << (n - 1).bit_length()
        az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64); az[:n] = a % np.uint64(BIGP); bz[:n] = b
        want = oracle.crt_ntt_multiply(az, bz, BIGP)[:2 * n - 1]
        got = emulated_fourstep_product(twin, a, b, BIGP, world, to_dev, from_dev, torch, None)
        assert np.array_equal(got, want), (world, n)
        if n in (3000, 20000, 70000):     # the loading pass without its shortcut for the padding rows
            got = emulated_fourstep_product(twin, a, b, BIGP, world, to_dev, from_dev, torch, None, all_rows=True)
            assert np.array_equal(got, want), (world, n, "all rows")


def check_one_column_slab_rows_pass(lib, to_dev, from_dev, stream, p=BIGP):
    """ntt64_fourstep_rows_device on slabs ONE column wide (column col0 of a 2^lr x 2^lc matrix) against the same thing
    assembled from its parts: the plain transform of the column followed by the explicit twiddle pass w^(col0 * k)."""
    rng = np.random.default_rng(61)
    for lr, lc, col0 in ((3, 3, 5), (5, 2, 3), (8, 4, 9), (13, 1, 1), (14, 3, 7), (1, 1, 1), (6, 6, 0)):
        lt = lr + lc
        x = rng.integers(0, p, 1 << lr, dtype=np.uint64)
        got = to_dev(x)
        lib.check(lib.lib.ntt64_fourstep_rows_device(got.data_ptr(), p, lt, lr, 1, col0, 0, stream))
        want = to_dev(x)
        lib.check(lib.lib.ntt64_transform_device(want.data_ptr(), p, lt, lr, 1, 1, 0, stream))
        lib.check(lib.lib.ntt64_twiddle_device(want.data_ptr(), p, lt, 1 << lr, 1, 0, col0, lr, 1, 0, stream))
        assert np.array_equal(from_dev(got) % np.uint64(p), from_dev(want) % np.uint64(p)), (lr, lc, col0)
        # and back: the inverse pass undoes it up to the factor 2^lr
        lib.check(lib.lib.ntt64_fourstep_rows_device(got.data_ptr(), p, lt, lr, 1, col0, 1, stream))
        back = from_dev(got).astype(object) % p * pow(1 << lr, p - 2, p) % p
        assert np.array_equal(back.astype(np.uint64), x), (lr, lc, col0)


def test_rows_pass_of_a_one_column_slab_keeps_its_twiddles(twin):
    import torch
    check_one_column_slab_rows_pass(twin, lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()),
                                    lambda t: t.numpy().view(np.uint64).copy(), None)


def test_library_workspaces_are_capped_and_can_be_released(twin):
    """NULL-workspace device calls keep one library-owned workspace per stream, at most eight of them; more streams evict
    the least recently used, and ntt_release_workspace frees them on request -- results stay right throughout."""
    import torch
    rng = np.random.default_rng(8)
    p, n = 998244353, 16384
    a = rng.integers(0, p, n, dtype=np.uint32); b = rng.integers(0, p, n, dtype=np.uint32)
    from oracle_binding import Oracle
    want = Oracle().ntt_multiply(a, b, p)
    ta, tb = torch.from_numpy(a.view(np.int32).copy()), torch.from_numpy(b.view(np.int32).copy())
    for fake_stream in list(range(1, 12)) + [3, 1]:       # the twin ignores the handle's meaning; the map is keyed by it
        out = torch.zeros(2 * n - 1, dtype=torch.int32)
        twin.check(twin.lib.ntt_poly_multiply_batched_device(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, p, 1, n, n, 2 * n - 1, None, 0, fake_stream))
        assert np.array_equal(out.numpy().view(np.uint32).astype(np.uint64), want)
    assert twin.lib.ntt_release_workspace(3, 0) == 0 and twin.lib.ntt_release_workspace(777, 0) == 0
    assert twin.lib.ntt_release_workspace(None, 1) == 0
    out = torch.zeros(2 * n - 1, dtype=torch.int32)
    twin.check(twin.lib.ntt_poly_multiply_batched_device(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, p, 1, n, n, 2 * n - 1, None, 0, 5))
    assert np.array_equal(out.numpy().view(np.uint32).astype(np.uint64), want)


def test_route_selection_beyond_the_reference_range(twin):
    """Lengths above 2^18 and large moduli above the CRT bound go to the 64-bit field (shift property, exact)."""
    rng = np.random.default_rng(18)
    for p, n in ((BIGP, 1 << 18), (998244353, 1 << 19)):
        a = rng.integers(0, p, n, dtype=np.uint64)
        e = np.zeros(n, dtype=np.uint64); e[5] = 1
        got = twin.multiply_u64(a, e, p)
        assert np.array_equal(got[5:5 + n], a) and not got[:5].any() and not got[5 + n:].any()
    assert twin.describe_plan(1 << 17, BIGP).startswith("k64_")
    assert twin.set_route_policy("reference") == "auto"
    assert twin.describe_plan(1 << 17, BIGP).endswith("k_inv_strided_crt")   # merge fused into the inverse column pass
    assert twin.describe_plan(1 << 18, BIGP).startswith("k64_")       # past the CRT route's length limit either way
    assert twin.set_route_policy("auto") == "reference"
    assert twin.lib.ntt_set_route_policy(7) == -1 and "route policy" in twin.last_error()
    assert twin.describe_plan(1 << 19, 998244353).startswith("k64_")


def test_standalone_transform_matches_reference_ntt(twin, oracle):
    """ntt_transform_u64 = NTT(a, n, invert, p, 3) of main_naive.cc:96-137 (natural order, inverse scaled by n^-1)."""
    rng = np.random.default_rng(96)
    for p in (7340033, 998244353, 1224736769):
        for n in (1, 2, 4, 64, 1024, 4096, 8192, 32768):
            x = rng.integers(0, p, n, dtype=np.uint64)
            fwd = twin.transform_u64(x, False, p)
            assert np.array_equal(fwd, oracle.ntt(x, False, p)), (p, n)
            assert np.array_equal(twin.transform_u64(x, True, p), oracle.ntt(x, True, p)), (p, n)
            assert np.array_equal(twin.transform_u64(fwd, True, p), x)
    from ntt_amd._binding import NttError
    with pytest.raises(NttError):
        twin.transform_u64(np.ones(12, dtype=np.uint64), False, 998244353)   # not a power of two


def test_natural_order_transform_on_32_bit_residues(twin, oracle):
    """the 32-bit transform kernels (ntt_transform32.h) for every length 2 .. 2^18 and both entry points; 257 and 65537
    have no 64th / short 2-power roots beyond their size and take the radix-2 kernel"""
    rng = np.random.default_rng(137)
    for p in (7340033, 998244353, 1224736769, 257, 65537):
        for lg in range(1, 19):
            n = 1 << lg
            if (p - 1) % n:
                continue
            batch = 3 if lg <= 13 else 1
            x = rng.integers(0, p, (batch, n), dtype=np.uint64)
            x[0, 0] = p - 1
            for inv in (0, 1):
                want = np.stack([oracle.ntt(x[r], bool(inv), p) for r in range(batch)])
                y = x.astype(np.uint32).copy()
                twin.check(twin.lib.ntt_transform_device_u32(y.ctypes.data, n, batch, inv, p, None))
                assert np.array_equal(y.astype(np.uint64), want), (p, lg, inv)
                z = x.copy()
                twin.check(twin.lib.ntt_transform_device_u64(z.ctypes.data, n, batch, inv, p, None))
                assert np.array_equal(z, want), (p, lg, inv)


def test_host_u64_path_narrows_reduced_inputs_and_reduces_the_rest(twin, oracle):
    # reduced inputs below a 31-bit modulus travel as 32-bit words; a single value >= p sends the call down the
    # 64-bit loads, which reduce it -- both must give the product of the residues
    rng = np.random.default_rng(12)
    for p, n in [(998244353, 1024), (1224736769, 4096), (7340033, 4)]:
        a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
        want = oracle.poly_multiply_u64(a, b, p)
        assert np.array_equal(twin.multiply_u64(a, b, p), want)
        a2 = a.copy(); a2[n // 2] += np.uint64(p) * np.uint64(3); b2 = b.copy(); b2[0] = np.uint64(2**64 - 1)
        want2 = oracle.poly_multiply_u64(a2 % np.uint64(p), b2 % np.uint64(p), p)
        assert np.array_equal(twin.multiply_u64(a2, b2, p), want2)
        ab = rng.integers(0, p, (3, n), dtype=np.uint64); bb = rng.integers(0, p, (3, n), dtype=np.uint64)
        got = twin.multiply_batched_u64(ab, bb, p)
        for r in range(3):
            assert np.array_equal(got[r], oracle.poly_multiply_u64(ab[r], bb[r], p))


def _bitrev_perm(bits):
    idx = np.arange(1 << bits)
    rev = np.zeros_like(idx)
    for b in range(bits):
        rev |= ((idx >> b) & 1) << (bits - 1 - b)
    return rev


def check_middle_axis_transform(lib, oracle, to_dev, from_dev, stream, log_root, log_len, inner, outer, p=998244353):
    """ntt64_transform_device along the middle axis of [outer][2^log_len][inner] against the reference transform
    applied to every (outer, inner) column; w = root of order 2^log_len derived from the plan's root.  The modulus
    is a 32-bit prime because the reference transform multiplies in 64 bits (main_naive.cc:116-121)."""
    rng = np.random.default_rng(log_len * 131 + inner)
    n = 1 << log_len
    x = rng.integers(0, p, (outer, n, inner), dtype=np.uint64)
    dev = to_dev(x)
    lib.check(lib.lib.ntt64_transform_device(dev.data_ptr(), p, log_root, log_len, inner, outer, 0, stream))
    got = from_dev(dev).reshape(outer, n, inner) % np.uint64(p)
    rev = _bitrev_perm(log_len)
    for o in range(outer):
        for c in range(inner):
            want = oracle.ntt(x[o, :, c], False, p)
            assert np.array_equal(got[o, rev, c], want), (log_len, inner, o, c)
    lib.check(lib.lib.ntt64_transform_device(dev.data_ptr(), p, log_root, log_len, inner, outer, 1, stream))
    back = from_dev(dev).reshape(outer, n, inner)
    ninv = pow(n, p - 2, p)
    assert np.array_equal((back.astype(object) * ninv % p).astype(np.uint64), x)


def test_transform_along_a_strided_axis_tiled_and_register_passes(twin, oracle):
    import torch
    to_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy())
    from_dev = lambda t: t.numpy().view(np.uint64).copy()
    # (log_len, inner, outer): odd and even inner extents, tiles of several shapes, register-pass fall-back
    for log_len, inner, outer in [(10, 6, 2), (9, 8, 1), (11, 16, 1), (5, 3, 3), (13, 1, 2), (14, 1, 1), (4, 40, 2)]:
        check_middle_axis_transform(twin, oracle, to_dev, from_dev, None, 20, log_len, inner, outer)


def test_wide_crt_route_any_modulus(twin, oracle):
    """Moduli and sizes that neither the four 31-bit primes nor the modulus itself can carry: two or three primes of 50
    bits (double-precision field: needs n*(p-1)^2 below 2^100 resp. 2^151) or of 59 bits (integer field: 2^117, 2^176) and
    Garner recombination.  Checked against the schoolbook product in 128-bit arithmetic."""
    rng = np.random.default_rng(59)
    cases = [(2**64 - 59, 300),            # largest 64-bit prime: three 50-bit primes
             (2**63 + 29, 64), (2**62 - 57, 1000),
             (2**61 - 1, 257),             # above the 64-bit field's limit
             (10**18 + 9, 100), (2**64 - 1, 50), (2**64 - 2**32, 33),   # composite and even moduli
             (2**60 + 33, 1)]
    for p, n in cases:
        a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
        a[0] = p - 1; b[-1] = p - 1
        assert twin.describe_plan(n, p).endswith("k64_garner"), (p, n)
        assert np.array_equal(twin.multiply_u64(a, b, p), oracle.schoolbook_u64(a, b, p)), (p, n)
    # extreme values, every coefficient p-1: the bound n*(p-1)^2 is met with equality in the middle coefficient
    p, n = 2**64 - 59, 512
    a = np.full(n, p - 1, dtype=np.uint64)
    assert np.array_equal(twin.multiply_u64(a, a, p), oracle.schoolbook_u64(a, a, p))
    # lengths above 2^18 with moduli that are not NTT-friendly: 121 bits -> three 50-bit primes; 115 bits -> two 59-bit
    # primes (three 50-bit ones would be more work); 85 bits -> two 50-bit primes
    for p in (1337006139375617 + 2, 2**47 + 5, 4294967311):
        n = 1 << 18
        a = rng.integers(0, p, n, dtype=np.uint64)
        e = np.zeros(n, dtype=np.uint64); e[3] = 2
        got = twin.multiply_u64(a, e, p)
        assert twin.describe_plan(n, p).endswith("k64_garner")
        assert np.array_equal(got[3:3 + n], (a.astype(object) * 2 % p).astype(np.uint64)) and not got[:3].any(), p
    # batched rows
    p, n = 2**63 + 29, 40
    ab = rng.integers(0, p, (3, n), dtype=np.uint64); bb = rng.integers(0, p, (3, n), dtype=np.uint64)
    got = twin.multiply_batched_u64(ab, bb, p)
    for r in range(3):
        assert np.array_equal(got[r], oracle.schoolbook_u64(ab[r], bb[r], p))


WIDE_INT_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntt_amd._binding import NttLibrary
from oracle_binding import Oracle
lib = NttLibrary(LIB); orc = Oracle()
rng = np.random.default_rng(176)
ok = True
for p, n in ((2**64 - 59, 300), (2**63 + 29, 64), (1000000007, 3000), (2**64 - 1, 50)):
    a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
    a[0] = p - 1; b[-1] = p - 1
    ok = ok and lib.describe_plan(n, p).endswith("k64_garner" if p > 2**32 else "k_crt_merge")
    ok = ok and np.array_equal(lib.multiply_u64(a, b, p), orc.schoolbook_u64(a, b, p))
print("RESULT", ok)
'''


def test_wide_crt_route_on_the_integer_basis(twin, tmp_path):
    """NTT_FIELD64=int keeps every product on Montgomery integers, so the same moduli go through the 59-bit primes
    (three of them for the 64-bit moduli: 137 bits)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "wide_int.py"
    script.write_text(f"ROOT = {root!r}\nLIB = {twin.path!r}\n" + WIDE_INT_WORKER)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, NTT_FIELD64="int"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True" in r.stdout, r.stdout[-500:]


def check_lifecycle_and_argument_errors(lib, oracle):
    """Shared by the CPU-twin and the GPU suite: argument checking, empty batches, untouched tail, shutdown and re-use."""
    import ctypes
    u64p = ctypes.POINTER(ctypes.c_uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    L = lib.lib
    p = 998244353
    a = np.arange(1, 9, dtype=np.uint64); b = np.arange(2, 10, dtype=np.uint64)
    # the slot after the 2n-1 results is never written (the reference's arrays are zeroed by its caller, main_naive.cc:250)
    out = np.full(16, 77, dtype=np.uint64)
    assert L.ntt_poly_multiply_u64(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p), 8, p) == 0
    assert np.array_equal(out[:15], oracle.ntt_multiply(a, b, p)) and out[15] == 77
    # argument errors: null pointers, n < 1, negative / zero modulus for the int entry point
    assert L.ntt_poly_multiply_u64(None, b.ctypes.data_as(u64p), out.ctypes.data_as(u64p), 8, p) == -1
    assert L.ntt_poly_multiply_u64(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p), 0, p) == -1
    i32p = ctypes.POINTER(ctypes.c_int32)
    ai = np.arange(4, dtype=np.int32); oi = np.zeros(8, dtype=np.int32)
    assert L.ntt_poly_multiply_i32(ai.ctypes.data_as(i32p), ai.ctypes.data_as(i32p), oi.ctypes.data_as(i32p), 4, -5) == -1
    assert "modulus" in lib.last_error()
    # an empty batch is a successful no-op
    z = np.zeros((0, 8), dtype=np.uint32)
    assert L.ntt_poly_multiply_batched(z.ctypes.data_as(u32p), z.ctypes.data_as(u32p), z.ctypes.data_as(u32p), 8, p, 0, 1) == 0
    # shutdown releases plans and buffers; the next call sets everything up again
    first = lib.multiply_u64(a, b, p)
    lib.shutdown()
    assert L.ntt_device_count() == 0
    again = lib.multiply_u64(a, b, p)
    assert np.array_equal(first, again) and L.ntt_device_count() >= 1
    big = 1337006139375617
    x = np.array([big - 1] * 4, dtype=np.uint64)
    assert np.array_equal(lib.multiply_u64(x, x, big), oracle.crt_ntt_multiply(x, x, big))


def test_lifecycle_and_argument_errors(twin, oracle):
    check_lifecycle_and_argument_errors(twin, oracle)


PIPELINED_WORKER = r'''
import os, sys
import os

import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntt_amd._binding import NttLibrary
from oracle_binding import Oracle
lib = NttLibrary(LIB); orc = Oracle()
rng = np.random.default_rng(8)
p, n, batch = 998244353, 2048, 43          # 1.4 MB of rows against a 1 MB limit: chunks of 8 rows, a ragged last one
a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
got = lib.multiply_batched(a, b, p, 1)
ok = all(np.array_equal(got[r].astype(np.uint64), orc.ntt_multiply(a[r], b[r], p)) for r in range(batch))
P, n2, batch2 = 1337006139375617, 1024, 21  # 64-bit rows through both large-modulus routes, workspace per staging set
A = rng.integers(0, P, (batch2, n2), dtype=np.uint64); B = rng.integers(0, P, (batch2, n2), dtype=np.uint64)
ok2 = True
for policy in ("auto", "reference"):
    lib.set_route_policy(policy)
    got2 = lib.multiply_batched_u64(A, B, P, 1)
    ok2 = ok2 and all(np.array_equal(got2[r], orc.crt_ntt_multiply(A[r], B[r], P)) for r in range(batch2))
one = lib.multiply_batched(a[:1], b[:1], p, 1)   # small call afterwards: single-buffer path again
print("RESULT", ok, ok2, bool(np.array_equal(one[0], got[0])))
'''


def run_pipelined_worker(lib_path, tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "pipelined.py"
    script.write_text(f"ROOT = {root!r}\nLIB = {lib_path!r}\n" + PIPELINED_WORKER)
    env = dict(os.environ, NTT_STAGE_LIMIT_MB="1")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True True True" in r.stdout, r.stdout[-500:]


def test_large_host_calls_go_through_in_chunks(twin, tmp_path):
    """Host-pointer calls beyond the staging limit (lowered here by the test hook) alternate between two staging sets."""
    run_pipelined_worker(twin.path, tmp_path)


TAIL_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntt_amd._binding import NttLibrary
from oracle_binding import Oracle
lib = NttLibrary(LIB); orc = Oracle()
rng = np.random.default_rng(11)
p, n = 998244353, 4096
ok = True
for batch in (31, 127, 257):
    a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
    got = lib.multiply_batched(a, b, p, 1)
    got64 = lib.multiply_batched_u64(a.astype(np.uint64), b.astype(np.uint64), p, 1)   # narrowed on the way in, widened on the way out
    for r in (0, batch - 2, batch - 1):
        want = orc.ntt_multiply(a[r], b[r], p)
        ok = ok and np.array_equal(got[r].astype(np.uint64), want) and np.array_equal(got64[r], want)
print("RESULT", ok)
'''


@pytest.mark.parametrize("workers,limit_mb", [(0, 64), (1, 64), (3, 64), (6, 64), (7, 64), (7, 1), (2, 2)])
def test_staged_copies_keep_the_tail_of_odd_sized_batches(twin, tmp_path, workers, limit_mb):
    """Row counts whose byte total is not a multiple of (copy threads x 64 B): the last coefficients of the last row must arrive
    (inputs) and come back (outputs) whatever the number of copy threads and the staging limit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tail.py"
    script.write_text(f"ROOT = {root!r}\nLIB = {twin.path!r}\n" + TAIL_WORKER)
    env = dict(os.environ, NTT_COPY_WORKERS=str(workers), NTT_STAGE_LIMIT_MB=str(limit_mb))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True" in r.stdout, r.stdout[-500:]


MULTI_DEVICE_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntt_amd._binding import NttLibrary, NttError
from oracle_binding import Oracle
lib = NttLibrary(LIB); orc = Oracle()
lib.init(0)
assert lib.lib.ntt_device_count() == 3
rng = np.random.default_rng(31)
p, P = 998244353, 1337006139375617
ok = True
for n, batch, gpus in ((2048, 7, 3), (2048, 7, 2), (512, 2, 3), (16384, 5, 3), (100, 1, 3), (4096, 9, 0)):
    a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
    got = lib.multiply_batched(a, b, p, gpus)
    got64 = lib.multiply_batched_u64(a.astype(np.uint64), b.astype(np.uint64), p, gpus)
    for r in range(batch):
        want = orc.ntt_multiply(a[r], b[r], p) if n & (n - 1) == 0 else orc.schoolbook_u64(a[r], b[r], p)
        ok = ok and np.array_equal(got[r].astype(np.uint64), want) and np.array_equal(got64[r], want)
A = rng.integers(0, P, (5, 1024), dtype=np.uint64); B = rng.integers(0, P, (5, 1024), dtype=np.uint64)
got = lib.multiply_batched_u64(A, B, P, 3)                      # four-prime route, shards of 2, 2, 1 rows
ok = ok and all(np.array_equal(got[r], orc.crt_ntt_multiply(A[r], B[r], P)) for r in range(5))
U = rng.integers(0, 2**64, (4, 256), dtype=np.uint64)            # unreduced values: not narrowed, reduced on the device
got = lib.multiply_batched_u64(U, U, p, 3)
ok = ok and all(np.array_equal(got[r], orc.ntt_multiply(U[r] % p, U[r] % p, p)) for r in range(4))
err = ""
try:
    lib.multiply_batched(A[:, :8].astype(np.uint32), A[:, :8].astype(np.uint32), 1000000007, 3)   # refused before any device work
except NttError as e:
    err = str(e)
print("RESULT", ok, "NTT-friendly" in err)
'''


def test_batches_sharded_over_several_devices_inside_the_library(twin, tmp_path):
    """ntt_poly_multiply_batched(_u64)(..., n_gpus): the stand-in reports three devices, so the per-device host threads, uneven
    shards (3 + 3 + 1, 2 + 2 + 1), batches smaller than the device count and the narrowing path all run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "multidev.py"
    script.write_text(f"ROOT = {root!r}\nLIB = {twin.path!r}\n" + MULTI_DEVICE_WORKER)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, NTT_TWIN_DEVICES="3"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True True" in r.stdout, r.stdout[-500:]


def test_short_products_several_per_workgroup(twin, oracle):
    """Transform lengths 32..4096 run 4096 / L products per workgroup (k_small16): batches that do not fill the last
    workgroup, every length class, both lazy and non-lazy primes; lengths below 32 keep the radix-2 kernel."""
    rng = np.random.default_rng(16)
    for p in (998244353, 1224736769, 7340033):
        for n, batch in ((16, 300), (17, 5), (31, 129), (100, 33), (256, 17), (500, 9), (1024, 5), (2048, 3), (2048, 1), (8, 1000), (3, 7)):
            a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
            a[0] = p - 1; b[0] = p - 1
            got = twin.multiply_batched(a, b, p, 1)
            for r in sorted({0, 1 % batch, batch // 2, batch - 1}):
                want = oracle.ntt_multiply(a[r], b[r], p) if n & (n - 1) == 0 else oracle.schoolbook_u64(a[r], b[r], p)
                assert np.array_equal(got[r].astype(np.uint64), want), (p, n, batch, r)
    assert twin.describe_plan(1000, 998244353) == "k_small16" and twin.describe_plan(8, 998244353) == "k_small"
