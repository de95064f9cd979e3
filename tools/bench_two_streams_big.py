#!/usr/bin/env python3
"""A/B: large-modulus products (n = 2^17, p = 19*2^46 + 1, 64-bit field route) as one stream of 128-product calls against
two / four HIP streams with 64- / 32-product calls each, so that one stream's memory-bound tile passes can overlap the
other's issue-bound core.  Steady state, products per second of the whole GPU."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library

P, N, B = 1337006139375617, 131072, 128


def main():
    lib = load_product_library(); lib.init(0)
    L = lib.lib
    a = torch.randint(0, P, (B, N), device="cuda", dtype=torch.int64)
    b = torch.randint(0, P, (B, N), device="cuda", dtype=torch.int64)
    out = torch.zeros((B, 2 * N - 1), dtype=torch.int64, device="cuda")

    def run(n_streams, steps):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        per = B // n_streams
        ws_bytes = L.ntt_workspace_bytes(N, P, per)
        ws = [torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") for _ in range(n_streams)]
        def step():
            for s, st in enumerate(streams):
                lo = s * per
                lib.check(L.ntt_poly_multiply_batched_device_u64(a[lo:].data_ptr(), b[lo:].data_ptr(), out[lo:].data_ptr(),
                                                                 N, P, per, N, N, 2 * N - 1, ws[s].data_ptr(), ws_bytes, st.cuda_stream))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:      # ramp
            for _ in range(4):
                step()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step()
        torch.cuda.synchronize()
        return B * steps / (time.perf_counter() - t0)

    for rnd in range(2):
        for ns in (1, 2, 4):
            print(json.dumps({"streams": ns, "products_per_call": B // ns, "round": rnd, "products_per_s": round(run(ns, 50))}), flush=True)


if __name__ == "__main__":
    main()
