#!/usr/bin/env python3
"""Times single passes of the 64-bit-field transform on a 2^29-word array through ntt64_transform_device, for a set of
(log_len, inner, outer) shapes: the effect of the row stride and of the tile shape on a pass, apart from everything else."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_amd._binding import load_product_library

P = 1337006139375617


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":
        from ntt_amd._binding import NttLibrary
        lib = NttLibrary(os.path.abspath(sys.argv[2])); del sys.argv[1:3]
    else:
        lib = load_product_library()
    lib.init(0)
    total = 29
    x = torch.randint(0, P, (1 << total,), device="cuda", dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or \
        [(9, 20, 0), (9, 11, 9), (9, 4, 16), (8, 12, 9), (8, 21, 0), (5, 24, 0), (7, 22, 0), (12, 0, 17), (6, 23, 0), (9, 15, 5)]
    for log_len, log_inner, log_outer in shapes:
        assert log_len + log_inner + log_outer == total
        def run(inv):
            lib.check(lib.lib.ntt64_transform_device(x.data_ptr(), P, total, log_len, 1 << log_inner, 1 << log_outer, inv, st))
        run(0); run(1); torch.cuda.synchronize()
        res = {}
        for inv in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(inv)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            res["inv" if inv else "fwd"] = round(ms, 3)
        print(json.dumps({"log_len": log_len, "row_stride_bytes": 8 << log_inner, "outer": 1 << log_outer, **res,
                          "TBps_fwd": round((16 << total) / res["fwd"] / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
