"""Arbitrary-modulus long products (wide CRT route): one product of n coefficients modulo p, operands resident in HBM.
usage: python tools/bench_wide_crt.py [log2 n] [p]      (NTT_FIELD64=int in the environment keeps the 59-bit integer basis)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ntt_amd._binding import load_product_library
lib = load_product_library(); lib.init(0)
L = lib.lib
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
p = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000007
n = 1 << logn
g = torch.Generator(device="cuda"); g.manual_seed(1)
a = torch.randint(0, min(p, 2**62), (n,), generator=g, device="cuda", dtype=torch.int64)
b = torch.randint(0, min(p, 2**62), (n,), generator=g, device="cuda", dtype=torch.int64)
out = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def step():
    lib.check(L.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, 1, n, n, 2 * n - 1, None, 0, st))
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"n=2^{logn} p={p} basis={'integer' if os.environ.get('NTT_FIELD64') == 'int' else 'auto'}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms  kernels: {lib.describe_plan(n, p)}")
