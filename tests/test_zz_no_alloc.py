"""The entry point that allocates nothing (ntt_poly_multiply_batched_device_noalloc) against the existing device entry point,
through the C ABI, bit for bit: two calls in a row, the second with a smaller input, so that anything the first call left
in the GPU's static scratch array would show.  Also its limits: an oversized batch and an unprepared (n, p) are refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 998244353


def _both(gpu_lib, torch, n, batch, seed, stride_pad=0):
    L = gpu_lib.lib
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    sa, so = n + stride_pad, 2 * n - 1 + stride_pad
    a = torch.randint(0, P, (batch, sa), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, P, (batch, sa), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    want = torch.full((batch, so), -1, dtype=torch.int32, device="cuda")
    got = torch.full((batch, so), -1, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ws_bytes = L.ntt_workspace_bytes(n, P, batch)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    gpu_lib.check(L.ntt_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), want.data_ptr(), n, P, batch, sa, sa, so,
                                                     ws.data_ptr(), ws_bytes, st))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    gpu_lib.check(L.ntt_poly_multiply_batched_device_noalloc(a.data_ptr(), b.data_ptr(), got.data_ptr(), n, P, batch, sa, sa, so, st))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated or freed during the call"
    return a, b, want, got


def test_noalloc_entry_point_matches_the_allocating_one(gpu_lib, oracle):
    import torch
    # first call: the largest batch of n = 2^17 products the 32 MiB scratch takes
    a, b, want, got = _both(gpu_lib, torch, 131072, 16, 1)
    assert torch.equal(got, want)
    ref = oracle.ntt_multiply(a[15].cpu().numpy().view(np.uint32), b[15].cpu().numpy().view(np.uint32), P)
    assert np.array_equal(got[15].cpu().numpy().view(np.uint32).astype(np.uint64), ref)
    # second call, smaller input, padded strides: same scratch, stale contents from the call before
    a, b, want, got = _both(gpu_lib, torch, 16384, 5, 2, stride_pad=3)
    assert torch.equal(got, want)
    assert bool((got[:, 2 * 16384 - 1:] == -1).all()), "wrote past 2n-1"
    # third: a single-kernel size (no scratch needed), any batch
    a, b, want, got = _both(gpu_lib, torch, 4096, 300, 3)
    assert torch.equal(got, want)


def test_noalloc_entry_point_refuses_what_it_cannot_serve(gpu_lib):
    import torch
    from ntt_amd._binding import NttError
    L = gpu_lib.lib
    n = 131072
    a = torch.zeros((17, n), dtype=torch.int32, device="cuda"); out = torch.zeros((17, 2 * n), dtype=torch.int32, device="cuda")
    ws_bytes = L.ntt_workspace_bytes(n, P, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    gpu_lib.check(L.ntt_poly_multiply_batched_device(a.data_ptr(), a.data_ptr(), out.data_ptr(), n, P, 1, n, n, 2 * n, ws.data_ptr(), ws_bytes, None))
    with pytest.raises(NttError, match="at most 16 products"):
        gpu_lib.check(L.ntt_poly_multiply_batched_device_noalloc(a.data_ptr(), a.data_ptr(), out.data_ptr(), n, P, 17, n, n, 2 * n, None))
    # a modulus no call of this process has prepared (167772161 = 5 * 2^25 + 1 appears in no other test)
    with pytest.raises(NttError, match="no cached tables"):
        gpu_lib.check(L.ntt_poly_multiply_batched_device_noalloc(a.data_ptr(), a.data_ptr(), out.data_ptr(), 3000, 167772161, 2, n, n, 2 * n, None))
    torch.cuda.synchronize()
