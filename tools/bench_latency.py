#!/usr/bin/env python3
"""Single-call latency of the host-pointer entry points (the number the reference prints as
"average latency"): wall time of one product including staging and PCIe, next to the same product
with operands resident in HBM.  usage: python tools/bench_latency.py [n ...]"""
import ctypes
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntt_amd._binding import load_product_library  # noqa: E402

u64p = ctypes.POINTER(ctypes.c_uint64)
i32p = ctypes.POINTER(ctypes.c_int32)


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[0], statistics.median(ts)


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [4, 4096, 131072]
    lib = load_product_library(); lib.init(1)
    L = lib.lib
    rng = np.random.default_rng(1)
    for p in (998244353, 1337006139375617):
        for n in sizes:
            a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
            out = np.zeros(2 * n - 1, dtype=np.uint64)
            pa, pb, po = a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p)
            f64 = lambda: L.ntt_poly_multiply_u64(pa, pb, po, n, p)
            for _ in range(5): f64()
            row = {"n": n, "p": p, "u64_host_us(min,med)": timed(f64, 200)}
            if p < 2**31:
                a32 = a.astype(np.int32); b32 = b.astype(np.int32); o32 = np.zeros(2 * n - 1, dtype=np.int32)
                qa, qb, qo = a32.ctypes.data_as(i32p), b32.ctypes.data_as(i32p), o32.ctypes.data_as(i32p)
                f32 = lambda: L.ntt_poly_multiply_i32(qa, qb, qo, n, p)
                for _ in range(5): f32()
                row["i32_host_us(min,med)"] = timed(f32, 200)
            da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
            do = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream

            def fdev():
                L.ntt_poly_multiply_batched_device_u64(da.data_ptr(), db.data_ptr(), do.data_ptr(), n, p, 1, n, n, 2 * n, None, 0, st)
                torch.cuda.synchronize()
            for _ in range(5): fdev()
            row["u64_device_resident_us(min,med)"] = timed(fdev, 200)
            # the same enqueue captured once into a graph and replayed
            ws_bytes = L.ntt_workspace_bytes(n, p, 1)
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
            side = torch.cuda.Stream()
            enq = lambda: L.ntt_poly_multiply_batched_device_u64(da.data_ptr(), db.data_ptr(), do.data_ptr(), n, p, 1, n, n, 2 * n,
                                                                 ws.data_ptr(), ws_bytes, side.cuda_stream)
            enq(); torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                enq()

            def fgraph():
                graph.replay(); torch.cuda.synchronize()
            for _ in range(5): fgraph()
            row["graph_replay_us(min,med)"] = timed(fgraph, 200)
            print(row, flush=True)


if __name__ == "__main__":
    main()
