"""Large-modulus products (n = 2^17, p = 1337006139375617) through the four-prime CRT route and through the
64-bit field directly, same inputs, steady state.  usage: python tools/bench_routes_big.py [--lib path.so] [batch] [n]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from ntt_amd._binding import load_product_library, NttLibrary
if len(sys.argv) > 2 and sys.argv[1] == "--lib":      # a diagnostic build
    lib = NttLibrary(os.path.abspath(sys.argv[2])); del sys.argv[1:3]
else:
    lib = load_product_library()
lib.init(0)
L = lib.lib
B, n, p = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 131072), 1337006139375617
rng = np.random.default_rng(1)
a = torch.from_numpy(rng.integers(0, p, (B, n), dtype=np.uint64).view(np.int64)).cuda()
b = torch.from_numpy(rng.integers(0, p, (B, n), dtype=np.uint64).view(np.int64)).cuda()
outs = [torch.zeros((B, 2 * n - 1), dtype=torch.int64, device="cuda") for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream
def crt():
    lib.check(L.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), outs[0].data_ptr(), n, p, B, n, n, 2 * n - 1, None, 0, st))
def field():
    lib.check(L.ntt64_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), outs[1].data_ptr(), n, p, B, n, n, 2 * n - 1, None, 0, st))
print("plan of the general entry point:", lib.describe_plan(n, p))
for name, fn in (("general entry point", crt), ("64-bit field", field)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn(); torch.cuda.synchronize()
    reps = 20 if B * n >= 1 << 20 else 200
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n} {name}: {dt*1e3:.3f} ms per batch of {B} -> {B/dt:.0f} products/s")
print("identical:", bool(torch.equal(outs[0], outs[1])))
