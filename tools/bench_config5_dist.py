#!/usr/bin/env python3
"""Config 5, multi-rank: one product of two length-n polynomials (default n = 2^28) modulo 19*2^46 + 1 by the
four-step transform with the matrix sharded over the ranks and an RCCL all-to-all between the two transform axes.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_config5_dist.py

Every rank generates its own column slabs on its GPU (a[i] = splitmix64(i) % p, b[i] = splitmix64(i + 2^32) % p), so
no 4 GiB input travels.  Rank 0 prints one JSON line; the check is A(1)*B(1) = C(1) and A(-1)*B(-1) = C(-1) mod p,
reduced over ranks."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from ntt_amd._binding import load_product_library
from ntt_amd.fourstep import FourStep

P = 1337006139375617


def splitmix64_mod(idx, p):
    """splitmix64 of an int64 index tensor (two's complement arithmetic), reduced mod p (< 2^63)"""
    def mul(x, c):  # wrapping 64-bit multiply by a constant given as unsigned
        c = c - (1 << 64) if c >= (1 << 63) else c
        return x * c
    def shr(x, s):  # logical shift right
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = idx + (0x9E3779B97F4A7C15 - (1 << 64))
    z = mul(z ^ shr(z, 30), 0xBF58476D1CE4E5B9)
    z = mul(z ^ shr(z, 27), 0x94D049BB133111EB)
    z = z ^ shr(z, 31)
    z = shr(z, 1)            # 63 significant bits: non-negative, so % is the mathematical remainder
    return z % p


def modsum(t, p):
    while t.numel() > 1:
        pad = (-t.numel()) % 1024
        if pad:
            t = torch.cat([t, torch.zeros(pad, dtype=t.dtype, device=t.device)])
        t = t.view(-1, 1024).sum(dim=1) % p
    return int(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=28)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    n = 1 << args.log_n
    world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # RCCL only: its transfers are ordered on the device stream, which the product relies on.  (gloo moves device
        # tensors from the host side without that ordering; the rank logic is rehearsed on CPU tensors in
        # tests/test_distributed.py and with emulated ranks on one GPU in tests/fourstep_emulation.py.)
        dist.init_process_group("nccl", device_id=dev)
    lib = load_product_library(); lib.init(0)
    fs = FourStep(lib, P, n, dev, dist if world > 1 else None)

    def slab(offset):
        r = torch.arange(fs.R, device=dev, dtype=torch.int64).view(-1, 1)
        j = torch.arange(fs.rank * fs.Cl, (fs.rank + 1) * fs.Cl, device=dev, dtype=torch.int64).view(1, -1)
        i = j + fs.C * r
        v = splitmix64_mod(i + offset, P)
        return torch.where(i < n, v, torch.zeros_like(v)).contiguous()

    a, b = slab(0), slab(1 << 32)
    def evals(t):  # (value at 1, value at -1) of this rank's share; column parity = index parity (C is even)
        cols = torch.arange(fs.rank * fs.Cl, (fs.rank + 1) * fs.Cl, device=dev) % 2
        ev, od = modsum(t[:, cols == 0].reshape(-1), P), modsum(t[:, cols == 1].reshape(-1), P)
        return torch.tensor([ev, od], dtype=torch.float64, device=dev), (ev, od)
    c = fs.multiply_slabs(a.clone(), b.clone())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        c = fs.multiply_slabs(a.clone(), b.clone())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.reps
    # verification through exact integer sums gathered on every rank
    def total(t):
        _, (ev, od) = evals(t)
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, (ev, od))
            ev, od = sum(x[0] for x in parts) % P, sum(x[1] for x in parts) % P
        return ev, od
    (ae, ao), (be, bo), (ce, co) = total(a), total(b), total(c)
    ok = ((ae + ao) * (be + bo) - (ce + co)) % P == 0 and ((ae - ao) * (be - bo) - (ce - co)) % P == 0
    if fs.rank == 0:
        print(json.dumps({"workload": f"config5 four-step n=2^{args.log_n} p={P} on {world} GPU(s), matrix {fs.R}x{fs.C}",
                          "ms": dt * 1e3, "all_to_alls_per_product": 3 if world > 1 else 0, "verified_x=+-1": bool(ok)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
