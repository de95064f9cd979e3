"""Config 3: 65536 products of n = 4096 coefficients (transform length 8192, one workgroup per product), operands in HBM.
usage: python tools/bench_config3.py [--lib path.so] [p]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ntt_amd._binding import load_product_library, NttLibrary
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    lib = NttLibrary(os.path.abspath(sys.argv[2])); del sys.argv[1:3]
else:
    lib = load_product_library()
lib.init(0)
p = int(sys.argv[1]) if len(sys.argv) > 1 else 998244353
n, B = 4096, 65536
a = torch.randint(0, p, (B, n), device="cuda", dtype=torch.int64).to(torch.int32)
b = torch.randint(0, p, (B, n), device="cuda", dtype=torch.int64).to(torch.int32)
out = torch.zeros((B, 2 * n - 1), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def step():
    lib.check(lib.lib.ntt_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, B, n, n, 2 * n - 1, None, 0, st))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"config 3 p={p}: {dt*1e3:.3f} ms per {B} products -> {B/dt/1e6:.2f} M products/s ({lib.describe_plan(n, p)})")
