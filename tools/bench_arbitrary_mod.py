#!/usr/bin/env python3
"""Batched products modulo a prime that is NOT NTT-friendly (1000000007): the four-prime route at short lengths --
one single-kernel product per prime and the recombination kernel -- products per second, operands resident in HBM.
usage: python tools/bench_arbitrary_mod.py [--lib path.so] [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library, NttLibrary

P = 1000000007


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":
        lib = NttLibrary(os.path.abspath(sys.argv[2])); del sys.argv[1:3]
    else:
        lib = load_product_library()
    lib.init(0)
    st = torch.cuda.current_stream().cuda_stream
    for n in [int(v) for v in sys.argv[1:]] or [64, 1024, 4096, 16384, 131072]:
        batch = max(1, (1 << 24) // n)
        a = torch.randint(0, P, (batch, n), device="cuda", dtype=torch.int64)
        b = torch.randint(0, P, (batch, n), device="cuda", dtype=torch.int64)
        out = torch.zeros((batch, 2 * n), dtype=torch.int64, device="cuda")
        run = lambda: lib.check(lib.lib.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, P, batch, n, n, 2 * n, None, 0, st))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            run(); torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"n={n} batch={batch} plan={lib.describe_plan(n, P)}: {dt * 1e3:.3f} ms -> {batch / dt:.0f} products/s", flush=True)


if __name__ == "__main__":
    main()
