"""Per-kernel times of the four-prime CRT product (n = 2^17, p = 1337006139375617) in steady state.
usage: python tools/profile_config4.py [--lib path.so] [batch] [auto|reference]   (route policy: direct 64-bit field / four-prime CRT)"""
import ctypes, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from ntt_amd._binding import load_product_library, NttLibrary
if len(sys.argv) > 2 and sys.argv[1] == "--lib":      # a diagnostic build (tools/ablate.sh)
    lib = NttLibrary(os.path.abspath(sys.argv[2])); del sys.argv[1:3]
else:
    lib = load_product_library()
lib.init(0)
lib.set_route_policy(sys.argv[2] if len(sys.argv) > 2 else "auto")
L = lib.lib
B, n, p = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 131072, 1337006139375617
rng = np.random.default_rng(1)
a = torch.from_numpy(rng.integers(0, p, (B, n), dtype=np.uint64).view(np.int64)).cuda()
b = torch.from_numpy(rng.integers(0, p, (B, n), dtype=np.uint64).view(np.int64)).cuda()
out = torch.zeros((B, 2 * n), dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def step():
    lib.check(L.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, B, n, n, 2 * n, None, 0, st))
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:   # past the clock transient
    step(); torch.cuda.synchronize()
print("kernels:", lib.describe_plan(n, p))
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"{dt*1e3:.3f} ms per batch of {B} -> {B/dt:.0f} products/s")
L.ntt_profile_enable(1)
for _ in range(5): step()
torch.cuda.synchronize()
names = ctypes.create_string_buffer(512); ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int * 16)()
L.ntt_profile_read.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
k = L.ntt_profile_read(names, 512, ms, cnt, 16)
for i, lab in enumerate(names.value.decode().split(",")):
    print(lab, "avg_us", round(ms[i] / cnt[i] * 1e3, 1), "launches/step", cnt[i] / 5, "us/step", round(ms[i] / 5 * 1e3, 1))
