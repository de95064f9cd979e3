"""GPU parity: the HIP library, called through its C ABI, against the CPU oracle on the same
seeded inputs -- bit-exact (integer arithmetic)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PRIMES32 = [7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769]
BIGP = 1337006139375617


def rand(rng, p, shape):
    return rng.integers(0, p, shape, dtype=np.uint64)


@pytest.mark.parametrize("p", PRIMES32)
@pytest.mark.parametrize("n", [1, 2, 4, 16, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072])
def test_direct_path_matches_oracle(gpu_lib, oracle, p, n):
    rng = np.random.default_rng(n * 7 + p % 1000)
    a, b = rand(rng, p, n), rand(rng, p, n)
    assert np.array_equal(gpu_lib.multiply_u64(a, b, p), oracle.ntt_multiply(a, b, p))


@pytest.mark.parametrize("n", [3, 5, 100, 1000, 3000, 4097, 5000, 100000])
def test_non_power_of_two_lengths(gpu_lib, oracle, n):
    # the reference needs n = 2^k; the oracle is fed the zero-extended operands instead
    p = 998244353
    rng = np.random.default_rng(n)
    a, b = rand(rng, p, n), rand(rng, p, n)
    m = 1 << (n - 1).bit_length()
    az = np.zeros(m, dtype=np.uint64); bz = np.zeros(m, dtype=np.uint64)
    az[:n] = a; bz[:n] = b
    want = oracle.ntt_multiply(az, bz, p)[:2 * n - 1]
    assert np.array_equal(gpu_lib.multiply_u64(a, b, p), want)


@pytest.mark.parametrize("p", [7340033, 998244353, 1224736769])
@pytest.mark.parametrize("n", [4, 4096, 131072])
def test_extreme_values(gpu_lib, oracle, p, n):
    # all coefficients p-1 (largest products), all zero, and a single one
    a = np.full(n, p - 1, dtype=np.uint64)
    assert np.array_equal(gpu_lib.multiply_u64(a, a, p), oracle.ntt_multiply(a, a, p))
    z = np.zeros(n, dtype=np.uint64)
    assert not gpu_lib.multiply_u64(a, z, p).any()
    e = z.copy(); e[n - 1] = 1
    got = gpu_lib.multiply_u64(a, e, p)
    assert np.array_equal(got[n - 1:], a) and not got[:n - 1].any()


@pytest.mark.parametrize("n", [1, 4, 1000, 4096, 8192, 131072])
def test_crt_path_matches_oracle(gpu_lib, oracle, n, route):
    rng = np.random.default_rng(n + 99)
    a, b = rand(rng, BIGP, n), rand(rng, BIGP, n)
    if n & (n - 1) == 0:
        want = oracle.crt_ntt_multiply(a, b, BIGP)
    else:
        want = oracle.schoolbook_u64(a, b, BIGP)
    assert np.array_equal(gpu_lib.multiply_u64(a, b, BIGP), want)


def test_crt_path_extreme_values(gpu_lib, oracle, route):
    n = 131072
    a = np.full(n, BIGP - 1, dtype=np.uint64)
    assert np.array_equal(gpu_lib.multiply_u64(a, a, BIGP), oracle.crt_ntt_multiply(a, a, BIGP))


@pytest.mark.parametrize("p,n", [(1000000007, 100), (97, 50), (2**31 - 1, 64), (4294967291, 16), (2**40 + 15, 2048)])
def test_unfriendly_moduli_take_the_crt_route(gpu_lib, oracle, p, n):
    rng = np.random.default_rng(5)
    a, b = rand(rng, p, n), rand(rng, p, n)
    assert np.array_equal(gpu_lib.multiply_u64(a, b, p), oracle.schoolbook_u64(a, b, p))


def test_i32_entry_point(gpu_lib, oracle):
    a = np.array([4, 1, 5, 2], dtype=np.int32); b = np.array([1, 5, 5, 4], dtype=np.int32)
    got = gpu_lib.multiply_i32(a, b, 7340033)
    assert got.tolist() == [4, 21, 30, 48, 39, 30, 8]          # nttdata/0.out
    assert np.array_equal(got, oracle.schoolbook_i32(a, b, 7340033))
    rng = np.random.default_rng(3)
    p = 998244353
    a = rng.integers(0, p, 512, dtype=np.int32); b = rng.integers(0, p, 512, dtype=np.int32)
    assert np.array_equal(gpu_lib.multiply_i32(a, b, p), oracle.schoolbook_i32(a, b, p))


@pytest.mark.parametrize("n,batch", [(4096, 64), (131072, 6), (1024, 33), (16384, 5)])
def test_batched_rows_are_independent_products(gpu_lib, oracle, n, batch):
    p = 998244353
    rng = np.random.default_rng(batch)
    a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
    got = gpu_lib.multiply_batched(a, b, p)
    for r in {0, 1, batch // 2, batch - 1}:
        assert np.array_equal(got[r].astype(np.uint64), oracle.ntt_multiply(a[r], b[r], p)), r


def test_batched_u64_crt(gpu_lib, oracle, route):
    n, batch = 4096, 5
    rng = np.random.default_rng(11)
    a, b = rand(rng, BIGP, (batch, n)), rand(rng, BIGP, (batch, n))
    got = gpu_lib.multiply_batched_u64(a, b, BIGP)
    for r in range(batch):
        assert np.array_equal(got[r], oracle.crt_ntt_multiply(a[r], b[r], BIGP)), r


def test_golden_files(gpu_lib, route):
    """Every committed golden vector: the reference's own data files (nttdata/0-3, filetest/0-3)
    and outputs of the compiled reference on seeded inputs (p = 998244353, CRT path, CRT primes)."""
    from golden_util import iter_golden
    seen = 0
    for name, n, p, a, b, out in iter_golden():
        assert np.array_equal(gpu_lib.multiply_u64(a, b, p), out), name
        seen += 1
    assert seen >= 16


def test_full_size_properties(gpu_lib):
    """n = 2^17 batch: linearity (a*(b+c) = a*b + a*c) and commutativity, checked on every coefficient."""
    p, n, batch = 998244353, 131072, 8
    rng = np.random.default_rng(2)
    a = rng.integers(0, p, (batch, n), dtype=np.uint32)
    b = rng.integers(0, p, (batch, n), dtype=np.uint32)
    c = rng.integers(0, p, (batch, n), dtype=np.uint32)
    bc = ((b.astype(np.uint64) + c) % p).astype(np.uint32)
    ab = gpu_lib.multiply_batched(a, b, p).astype(np.uint64)
    ac = gpu_lib.multiply_batched(a, c, p).astype(np.uint64)
    abc = gpu_lib.multiply_batched(a, bc, p).astype(np.uint64)
    assert np.array_equal(abc, (ab + ac) % p)
    assert np.array_equal(gpu_lib.multiply_batched(b, a, p).astype(np.uint64), ab)
    assert ab.max() < p


def test_device_resident_entry_point_with_strides(gpu_lib, oracle):
    import torch
    p, n, batch = 998244353, 16384, 3
    L = 2 * n
    rng = np.random.default_rng(8)
    a = rng.integers(0, p, (batch, n), dtype=np.int64); b = rng.integers(0, p, (batch, n), dtype=np.int64)
    da = torch.zeros((batch, n + 64), dtype=torch.int32, device="cuda"); da[:, :n] = torch.from_numpy(a).to("cuda", torch.int32)
    db = torch.from_numpy(b).to("cuda", torch.int32).contiguous()
    out = torch.full((batch, L), -1, dtype=torch.int32, device="cuda")
    st = gpu_lib.lib.ntt_poly_multiply_batched_device(da.data_ptr(), db.data_ptr(), out.data_ptr(), n, p, batch, n + 64, n, L,
                                                      None, 0, torch.cuda.current_stream().cuda_stream)
    gpu_lib.check(st)
    torch.cuda.synchronize()
    res = out.cpu().numpy()
    for r in range(batch):
        assert np.array_equal(res[r, :2 * n - 1].astype(np.uint64), oracle.ntt_multiply(a[r], b[r], p))
        assert res[r, 2 * n - 1] == -1   # index 2n-1 is never written (main_naive.cc:171-173 copies 2n-1 values)


def test_errors_are_reported_not_computed(gpu_lib):
    from ntt_amd._binding import NttError
    one = np.ones(4, dtype=np.uint64)
    with pytest.raises(NttError):
        gpu_lib.multiply_u64(one, one, 1)              # no modulus below 2
    with pytest.raises(NttError):
        gpu_lib.multiply_batched(np.ones((2, 8), dtype=np.uint32), np.ones((2, 8), dtype=np.uint32), 1000000007)


def test_routes_beyond_the_reference_range(gpu_lib, oracle):
    """n = 2^18 with the 51-bit modulus (past the four-prime CRT bound) and n = 2^19 with a 32-bit prime (past the
    2^18 transform limit of the 32-bit kernels) are served by the 64-bit field through the same entry point."""
    rng = np.random.default_rng(18)
    for p, n in ((BIGP, 1 << 18), (998244353, 1 << 19)):
        a = rng.integers(0, p, n, dtype=np.uint64)
        e = np.zeros(n, dtype=np.uint64); e[5] = 1
        got = gpu_lib.multiply_u64(a, e, p)
        assert np.array_equal(got[5:5 + n], a) and not got[:5].any() and not got[5 + n:].any()
    # and agree with the 32-bit route where both exist: n = 2^17 computed as a 2^18-length zero-extended operand
    p, n = 998244353, 1 << 17
    a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
    az = np.zeros(2 * n, dtype=np.uint64); bz = np.zeros(2 * n, dtype=np.uint64); az[:n] = a; bz[:n] = b
    assert np.array_equal(gpu_lib.multiply_u64(az, bz, p)[:2 * n - 1], oracle.ntt_multiply(a, b, p))


def test_host_u64_path_narrows_reduced_inputs_and_reduces_the_rest(gpu_lib, oracle):
    rng = np.random.default_rng(12)
    for p, n in [(998244353, 131072), (1224736769, 4096), (7340033, 4), (469762049, 65536)]:
        a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
        want = oracle.poly_multiply_u64(a, b, p)
        assert np.array_equal(gpu_lib.multiply_u64(a, b, p), want)
        a2 = a.copy(); a2[n // 2] += np.uint64(p) * np.uint64(3); b2 = b.copy(); b2[0] = np.uint64(2**64 - 1)
        want2 = oracle.poly_multiply_u64(a2 % np.uint64(p), b2 % np.uint64(p), p)
        assert np.array_equal(gpu_lib.multiply_u64(a2, b2, p), want2)
    # a call too large for the page-locked staging buffer takes the pageable path
    p, n, batch = 998244353, 4096, 2100
    a = rng.integers(0, p, (batch, n), dtype=np.uint64); b = rng.integers(0, p, (batch, n), dtype=np.uint64)
    got = gpu_lib.multiply_batched_u64(a, b, p)
    for r in (0, 1, batch // 2, batch - 1):
        assert np.array_equal(got[r], oracle.poly_multiply_u64(a[r], b[r], p))


def test_device_entry_point_can_be_captured_in_a_hip_graph(gpu_lib, oracle):
    # after one warm-up call (plan tables uploaded) and with a caller workspace the device entry point only
    # enqueues kernels, so a caller may capture it into a graph and replay it on new data
    import torch
    p, n, batch = 998244353, 16384, 3
    rng = np.random.default_rng(5)
    L = gpu_lib.lib
    a = torch.zeros((batch, n), dtype=torch.int32, device="cuda"); b = torch.zeros_like(a)
    out = torch.zeros((batch, 2 * n), dtype=torch.int32, device="cuda")
    ws_bytes = L.ntt_workspace_bytes(n, p, batch)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()

    def enqueue(stream):
        gpu_lib.check(L.ntt_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, batch, n, n, 2 * n,
                                                          ws.data_ptr(), ws_bytes, stream.cuda_stream))
    with torch.cuda.stream(side):
        enqueue(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        enqueue(side)
    for seed in (1, 2):
        ha = rng.integers(0, p, (batch, n), dtype=np.uint32); hb = rng.integers(0, p, (batch, n), dtype=np.uint32)
        a.copy_(torch.from_numpy(ha.view(np.int32))); b.copy_(torch.from_numpy(hb.view(np.int32)))
        out.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint32)
        for r in range(batch):
            assert np.array_equal(res[r, :2 * n - 1].astype(np.uint64), oracle.ntt_multiply(ha[r], hb[r], p))


def test_wide_crt_route_any_modulus(gpu_lib, oracle):
    """Same cases as the CPU-twin test of this name, on the GPU, plus a long product that only this route serves
    (1000000007 is not NTT-friendly and 2^19 coefficients overflow the four-prime bound), cross-checked against
    the 64-bit-field route on a modulus both can carry."""
    from test_cpu_twin import test_wide_crt_route_any_modulus as shared
    shared(gpu_lib, oracle)
    rng = np.random.default_rng(77)
    p, n = 1000000007, 1 << 19
    a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
    assert gpu_lib.describe_plan(n, p).endswith("k64_garner")
    got = gpu_lib.multiply_u64(a, b, p)
    ev = lambda v, s: (int(v[0::2].astype(object).sum()) + s * int(v[1::2].astype(object).sum())) % p
    for s in (1, -1):
        assert ev(a, s) * ev(b, s) % p == ev(got, s)
    assert int(got.max()) < p
    # head and tail coefficients against direct sums
    k = 5
    assert int(got[k]) == sum(int(a[i]) * int(b[k - i]) for i in range(k + 1)) % p
    assert int(got[2 * n - 2]) == int(a[-1]) * int(b[-1]) % p


def test_wide_crt_route_on_the_integer_basis(gpu_lib, tmp_path):
    """The same moduli with NTT_FIELD64=int (fresh process): every product in Montgomery integers, 59-bit primes."""
    import subprocess
    import sys
    from test_cpu_twin import WIDE_INT_WORKER
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "wide_int.py"
    script.write_text(f"ROOT = {root!r}\nLIB = {gpu_lib.path!r}\n" + WIDE_INT_WORKER)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, NTT_FIELD64="int"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True" in r.stdout, r.stdout[-500:]


def test_lifecycle_and_argument_errors(gpu_lib, oracle):
    from test_cpu_twin import check_lifecycle_and_argument_errors
    check_lifecycle_and_argument_errors(gpu_lib, oracle)


def test_large_host_calls_go_through_in_chunks(gpu_lib, tmp_path):
    from test_cpu_twin import run_pipelined_worker
    run_pipelined_worker(gpu_lib.path, tmp_path)


def test_crt_merge_against_python_integers(gpu_lib):
    """the recombination kernel alone (main_naive.cc:216-223): Garner digits and both tails -- exact doubles for final moduli
    up to 1.5e15, 128-bit integers above -- against Python integers, corners of the ranges included"""
    import torch
    import crt_merge_cases as cm
    rng = np.random.default_rng(216)
    xs = cm.cases(rng, 20000)

    def to_dev(arr):
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int32 if arr.dtype == np.uint32 else np.int64)).cuda()
        return {"t": t, "ptr": t.data_ptr(), "dtype": arr.dtype}

    def from_dev(d):
        torch.cuda.synchronize()
        return d["t"].cpu().numpy().view(d["dtype"])

    for p in cm.MODULI:
        cm.check(gpu_lib, to_dev, from_dev, p, xs)


@pytest.mark.parametrize("p", PRIMES32)
def test_natural_order_transform_on_32_bit_residues(gpu_lib, oracle, p):
    """ntt_transform_device_u32 / _u64 = the reference's NTT(a, n, invert, p, 3) (main_naive.cc:96-137: bit reversal, radix-2
    stages, n^-1 scaling of the inverse) for every length 2 .. 2^18 that divides p - 1, several rows per call, both
    directions, and forward followed by inverse gives the input back."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(p % 1009)
    for lg in range(1, 19):
        n = 1 << lg
        if (p - 1) % n:
            continue
        batch = 5 if lg <= 12 else (3 if lg <= 14 else 2)
        x = rand(rng, p, (batch, n))
        x[0, :] = p - 1                                      # the largest residues exercise the lazy ranges
        for inv in (0, 1):
            want = np.stack([oracle.ntt(x[r], bool(inv), p) for r in range(batch)])
            t32 = torch.from_numpy(x.astype(np.uint32).view(np.int32)).cuda()
            gpu_lib.check(gpu_lib.lib.ntt_transform_device_u32(t32.data_ptr(), n, batch, inv, p, st))
            t64 = torch.from_numpy(x.view(np.int64).copy()).cuda()
            gpu_lib.check(gpu_lib.lib.ntt_transform_device_u64(t64.data_ptr(), n, batch, inv, p, st))
            torch.cuda.synchronize()
            assert np.array_equal(t32.cpu().numpy().view(np.uint32).astype(np.uint64), want), (lg, inv)
            assert np.array_equal(t64.cpu().numpy().view(np.uint64), want), (lg, inv)
            if not inv:      # and back
                gpu_lib.check(gpu_lib.lib.ntt_transform_device_u32(t32.data_ptr(), n, batch, 1, p, st))
                torch.cuda.synchronize()
                assert np.array_equal(t32.cpu().numpy().view(np.uint32).astype(np.uint64), x), lg


def test_natural_order_transform_argument_errors(gpu_lib):
    import torch
    from ntt_amd._binding import NttError
    t = torch.zeros(64, dtype=torch.int32, device="cuda")
    for n, p in ((48, 998244353), (64, 1000000007), (1 << 19, 998244353)):      # not a power of two / unfriendly modulus / too long
        with pytest.raises(NttError):
            gpu_lib.check(gpu_lib.lib.ntt_transform_device_u32(t.data_ptr(), n, 1, 0, p, None))


@pytest.mark.parametrize("p", [7340033, 998244353, 1224736769])
def test_short_products_exchanged_inside_the_quad(gpu_lib, oracle, p):
    """transform lengths 32 and 64 (n = 9 .. 32): the last levels, the pointwise product and the first inverse levels run on
    registers exchanged between neighbouring lanes with DPP permutes (the butterflies of main.cc:926-995 at lane distance 1
    and 2); several products per workgroup, batches that do not fill the last workgroup"""
    rng = np.random.default_rng(p % 89)
    for n in (9, 12, 16, 17, 24, 31, 32):
        for batch in (1, 3, 130, 4099):
            a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
            a[0] = p - 1; b[0] = p - 1
            got = gpu_lib.multiply_batched(a, b, p)
            pick = sorted(set([0, 1, batch - 1] + list(range(0, batch, 97))))
            pick = [r for r in pick if r < batch]
            for r in pick:
                assert np.array_equal(got[r].astype(np.uint64), oracle.schoolbook_u64(a[r].astype(np.uint64), b[r].astype(np.uint64), p)), (n, batch, r)
