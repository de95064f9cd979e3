#!/usr/bin/env python3
"""A/B: the headline workload (n = 2^17, p = 998244353) as one stream of 256-product calls against two HIP streams that
each run 128-product calls, so that one stream's memory-bound column passes can overlap the other's issue-bound core.
Steady state (ramp first), distinct operand sets, products per second of the whole GPU."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library

P, N = 998244353, 131072


def main():
    lib = load_product_library(); lib.init(0)
    L = lib.lib
    sets = 4
    a = [torch.randint(0, P, (256, N), device="cuda", dtype=torch.int64).to(torch.int32) for _ in range(sets)]
    b = [torch.randint(0, P, (256, N), device="cuda", dtype=torch.int64).to(torch.int32) for _ in range(sets)]
    out = torch.zeros((256, 2 * N), dtype=torch.int32, device="cuda")

    def run(n_streams, steps):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        per = 256 // n_streams
        ws_bytes = L.ntt_workspace_bytes(N, P, per)
        ws = [torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") for _ in range(n_streams)]
        def step(k):
            for s, st in enumerate(streams):
                lo = s * per
                lib.check(L.ntt_poly_multiply_batched_device(a[k % sets][lo:].data_ptr(), b[k % sets][lo:].data_ptr(), out[lo:].data_ptr(),
                                                             N, P, per, N, N, 2 * N, ws[s].data_ptr(), ws_bytes, st.cuda_stream))
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.2:      # ramp
            for _ in range(8):
                step(k); k += 1
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(k + i)
        torch.cuda.synchronize()
        return 256 * steps / (time.perf_counter() - t0)

    for rnd in range(3):
        for ns in (1, 2, 4):
            print(json.dumps({"streams": ns, "products_per_call": 256 // ns, "round": rnd, "products_per_s": round(run(ns, 200))}), flush=True)


if __name__ == "__main__":
    main()
