#!/usr/bin/env python3
"""Products per second of short batched products (single-kernel paths), operands resident in HBM."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library

P = 998244353


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":
        from ntt_amd._binding import NttLibrary
        lib = NttLibrary(os.path.abspath(sys.argv[2])); del sys.argv[1:3]
    else:
        lib = load_product_library()
    lib.init(0)
    st = torch.cuda.current_stream().cuda_stream
    for n in [int(v) for v in sys.argv[1:]] or [64, 256, 512, 1024, 2048, 4096]:
        batch = max(1024, (1 << 28) // n)
        a = torch.randint(0, P, (batch, n), device="cuda", dtype=torch.int64).to(torch.int32)
        b = torch.randint(0, P, (batch, n), device="cuda", dtype=torch.int64).to(torch.int32)
        out = torch.zeros((batch, 2 * n), dtype=torch.int32, device="cuda")
        run = lambda: lib.check(lib.lib.ntt_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, P, batch, n, n, 2 * n, None, 0, st))
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(json.dumps({"n": n, "batch": batch, "kernel": lib.describe_plan(n, P), "ms": round(ms, 3), "products_per_s": round(batch / ms * 1e3),
                          "TBps_16n": round(batch * 16 * n / ms / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
