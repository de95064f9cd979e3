#!/usr/bin/env python3
"""Stand-alone natural-order transforms (the reference's NTT(), main_naive.cc:96-137), values per second and bytes moved
per value, device-resident rows: the 32-bit kernels through ntt_transform_device_u32 and through the 64-bit-word entry
point, against the 64-bit-field route of the same entry point (NTT_TRANSFORM32=off, run in a second process).
usage: python tools/bench_transform.py [p] [log2 n ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library


def main():
    args = sys.argv[1:]
    p = int(args[0]) if args else 998244353
    logs = [int(v) for v in args[1:]] or [10, 12, 13, 14, 16, 18]
    lib = load_product_library(); lib.init(0)
    st = torch.cuda.current_stream().cuda_stream
    route = "64-bit field" if os.environ.get("NTT_TRANSFORM32") == "off" else "32-bit kernels"
    for lg in logs:
        n = 1 << lg
        batch = max(1, (1 << 26) >> lg)          # 2^26 values per call
        x32 = torch.randint(0, p, (batch, n), device="cuda", dtype=torch.int64).to(torch.int32)
        x64 = x32.to(torch.int64)
        for name, fn in (("u32", lambda inv: lib.lib.ntt_transform_device_u32(x32.data_ptr(), n, batch, inv, p, st)),
                         ("u64", lambda inv: lib.lib.ntt_transform_device_u64(x64.data_ptr(), n, batch, inv, p, st))):
            if name == "u32" and route != "32-bit kernels":
                continue
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2:
                lib.check(fn(0)); lib.check(fn(1)); torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.check(fn(0)); lib.check(fn(1))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (2 * reps)
            print(json.dumps({"entry": name, "route": route, "p": p, "log2_n": lg, "batch": batch, "ms_per_transform_call": round(dt * 1e3, 4),
                              "values_per_s": round(batch * n / dt), "transforms_per_s": round(batch / dt)}), flush=True)


if __name__ == "__main__":
    main()
