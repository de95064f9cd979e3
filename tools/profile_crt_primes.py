import ctypes, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ntt_amd._binding import load_product_library
lib = load_product_library(); lib.init(0); L = lib.lib
B, n, p = 256, 131072, 1337006139375617
rng = np.random.default_rng(1)
a = torch.from_numpy(rng.integers(0, p, (B, n), dtype=np.uint64).view(np.int64)).cuda()
b = torch.from_numpy(rng.integers(0, p, (B, n), dtype=np.uint64).view(np.int64)).cuda()
res = torch.zeros((B, 2 * n), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L.ntt_profile_read.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
for prime in range(4):
    def step():
        lib.check(L.ntt_crt_residue_product_device(a.data_ptr(), b.data_ptr(), res.data_ptr(), n, prime, B, n, n, 2 * n, None, 0, st))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        step(); torch.cuda.synchronize()
    L.ntt_profile_enable(1)
    for _ in range(10): step()
    torch.cuda.synchronize()
    names = ctypes.create_string_buffer(512); ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int * 16)()
    k = L.ntt_profile_read(names, 512, ms, cnt, 16)
    print(L.ntt_crt_prime(prime), {lab: round(ms[i] / cnt[i] * 1e3, 1) for i, lab in enumerate(names.value.decode().split(","))})
    L.ntt_profile_enable(0)
