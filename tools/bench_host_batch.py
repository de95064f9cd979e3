#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer batched entry point (ntt_poly_multiply_batched): operands and results in
ordinary host memory, one call per batch.  usage: python tools/bench_host_batch.py [batch ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntt_amd._binding import load_product_library  # noqa: E402

lib = load_product_library(); lib.init(1)
n, p = 131072, 998244353
rng = np.random.default_rng(3)
for batch in [int(x) for x in sys.argv[1:]] or [8, 64, 256]:
    a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
    lib.multiply_batched(a, b, p, 1)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = lib.multiply_batched(a, b, p, 1)
    dt = (time.perf_counter() - t0) / reps
    gb = (a.nbytes + b.nbytes + out.nbytes) / 1e9
    print({"batch": batch, "n": n, "ms": round(dt * 1e3, 2), "products_per_s": round(batch / dt), "host_GBps": round(gb / dt, 2)}, flush=True)
