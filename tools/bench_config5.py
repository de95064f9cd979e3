#!/usr/bin/env python3
"""Config 5 on ONE GPU: a single product of two length-n polynomials (default n = 2^28, transform length 2^29)
modulo 1337006139375617 = 19*2^46 + 1 in the 64-bit field, (a) by the library's direct chain of passes and
(b) by the explicit four-step matrix form that the multi-rank version shards.  Operands are generated on the
device.  Checks: x = 1 and x = -1 evaluations and a monomial shift."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library

P = 1337006139375617


def modsum(t, p):
    """sum of a non-negative int64 tensor (entries < p < 2^51) modulo p, without overflow"""
    while t.numel() > 1:
        pad = (-t.numel()) % 1024
        if pad:
            t = torch.cat([t, torch.zeros(pad, dtype=t.dtype, device=t.device)])
        t = t.view(-1, 1024).sum(dim=1) % p
    return int(t.item())


def evaluate(t, sign, p):
    return (modsum(t[0::2], p) + sign * modsum(t[1::2], p)) % p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=28)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--direct-only", action="store_true")
    ap.add_argument("--warm", type=int, default=4, help="untimed products before the timed ones")
    ap.add_argument("--lib", default=None, help="a diagnostic build of the library (tools/ablate.sh) instead of the shipped one")
    args = ap.parse_args()
    n = 1 << args.log_n
    if args.lib:
        from ntt_amd._binding import NttLibrary
        lib = NttLibrary(os.path.abspath(args.lib))
    else:
        lib = load_product_library()
    lib.init(0)
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    a = torch.randint(0, P, (n,), generator=gen, device="cuda", dtype=torch.int64)
    b = torch.randint(0, P, (n,), generator=gen, device="cuda", dtype=torch.int64)
    out = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
    ws_bytes = lib.lib.ntt64_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")

    def direct():
        lib.check(lib.lib.ntt64_poly_multiply_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, P, ws.data_ptr(), ws_bytes, st))
    for _ in range(args.warm):     # the GPU's clocks dip during the first tens of milliseconds of a sustained load
        direct()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        direct()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    ok = all(evaluate(a, s, P) * evaluate(b, s, P) % P == evaluate(out, s, P) for s in (1, -1))
    ok = ok and int(out.max().item()) < P and int(out.min().item()) >= 0
    L = 2 << args.log_n
    print(json.dumps({"workload": f"config5 single GPU direct, n=2^{args.log_n}, p={P}", "ms": dt * 1e3,
                      "butterflies_per_s": 3 * (L // 2) * (args.log_n + 1) / dt, "verified_x=+-1": bool(ok)}), flush=True)

    if args.direct_only:
        return
    # explicit four-step on the same operands (world = 1): slabs built on the device
    from ntt_amd.fourstep import FourStep
    fs = FourStep(lib, P, n, torch.device("cuda", 0))
    def slab(v):
        z = torch.zeros(fs.R * fs.C, dtype=torch.int64, device="cuda"); z[:n] = v
        return z.view(fs.R, fs.C)
    for mode in ("fused", "stepwise"):
        times = []
        for rep in range(3):                 # the slabs are consumed: rebuild them outside the clock
            sa, sb = slab(a), slab(b)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c = fs.multiply_slabs(sa, sb) if mode == "fused" else fs.multiply_slabs_stepwise(sa, sb)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
            del sa, sb
        same = bool(torch.equal(c.view(-1)[: 2 * n - 1], out))
        print(json.dumps({"workload": f"config5 single GPU explicit four-step {fs.R}x{fs.C}, {mode}", "ms": min(times) * 1e3,
                          "ms_all": [round(x * 1e3, 2) for x in times], "equals_direct": same}), flush=True)
        del c


if __name__ == "__main__":
    main()
