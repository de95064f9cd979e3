#!/usr/bin/env python3
"""Headline benchmark: batched polynomial products at n = 2^17 modulo 998244353.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the multiply path over one batch of synthetic products whose inputs
and outputs are resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank per
GPU) the batch is sharded by product with no data-path collective; the barrier and the
max-over-ranks clock are the only communication.  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def cpu_baseline(n, p, min_seconds=12.0, max_reps=100000):
    """The reference's OpenMP multiply (compiled from its own source into oracle/_ref) timed on this
    host; falls back to this repo's port of the same algorithm when the reference build is absent."""
    import numpy as np
    from oracle_binding import Oracle, Reference
    orc = Oracle()
    a = orc.fill_random(n, p, 0x5EED0000); b = orc.fill_random(n, p, 0x5EED0001)
    ref = Reference()
    threads = 8  # NUM_THREADS of main_openMP.cc:15
    if ref.openmp is not None:
        kind, fn = "reference", (lambda: ref.openmp_ntt_multiply(a, b, p, threads))
    else:
        kind, fn = "port", (lambda: orc.ntt_multiply_openmp(a, b, p, threads))
    want = orc.ntt_multiply(a, b, p)
    got = fn(); fn()                      # warm-up (thread team start-up dominates the first call)
    assert np.array_equal(got, want)
    def rate(f, seconds):
        reps, t0 = 0, time.perf_counter()
        while True:
            f(); reps += 1
            el = time.perf_counter() - t0
            if el >= seconds or reps >= max_reps:
                return reps, el
    reps, el = rate(fn, min_seconds)
    out = {"value": reps / el, "unit": "poly-mults/s", "cores": threads, "kind": kind,
           "sample": f"{reps} products n={n} p={p} in {el:.1f} s, NTT_multiply_parallel (main_openMP.cc:176-217), {threads} OpenMP threads",
           "host_cpus": os.cpu_count()}
    # the same routine with one thread per core of this box's CPU share (the reference hard-codes 8, main_openMP.cc:15)
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 8)
    if share != threads:
        if ref.openmp is not None:
            fn2 = lambda: ref.openmp_ntt_multiply(a, b, p, share)
        else:
            fn2 = lambda: orc.ntt_multiply_openmp(a, b, p, share)
        fn2()
        reps2, el2 = rate(fn2, min_seconds / 2)
        out["value_all_cores"] = reps2 / el2
        out["cores_all"] = share
    return out


def load_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/traffic_*.json), which was
    collected at the batch size recorded in it (256 when absent); traffic is proportional to the batch."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        per_launch = d.get(kernel, {}).get("hbm_bytes_per_launch")
        return None if per_launch is None else per_launch * batch / d.get("_meta", {}).get("batch", 256)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed run of the same workload before the warm-up steps, long enough for the GPU's clock "
                         "management to settle (it dips for the first ~25 ms of a sustained load); 0 disables")
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--p", type=int, default=998244353)
    ap.add_argument("--batch", type=int, default=256, help="products per GPU per step")
    ap.add_argument("--input-sets", type=int, default=4,
                    help="distinct operand sets the steps rotate through, so that no step finds its inputs in the 256 MB "
                         "Infinity Cache from the step before")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of the library instead of the shipped one")
    ap.add_argument("--no-verify", action="store_true", help="diagnostic: skip the oracle spot-check (ablated builds)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from ntt_amd._binding import load_product_library

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    device_index = local_rank % torch.cuda.device_count()  # one rank per GPU; wraps only in single-GPU rehearsals
    torch.cuda.set_device(device_index)
    backend = None
    if world > 1:
        # RCCL carries the barrier and the max-reduce of the clock (the products themselves need no collective);
        # NTT_BENCH_BACKEND=gloo rehearses the multi-rank logic on a box with fewer GPUs than ranks
        backend = os.environ.get("NTT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    n, p, B = args.n, args.p, args.batch
    L = 2 * n
    if args.lib:
        from ntt_amd._binding import NttLibrary
        lib = NttLibrary(os.path.abspath(args.lib))
    else:
        lib = load_product_library()
    lib.init(0)

    gen = torch.Generator(device="cuda"); gen.manual_seed(0x5EED0000 + rank)
    sets = max(1, args.input_sets)
    a_sets = [torch.randint(0, p, (B, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32) for _ in range(sets)]
    b_sets = [torch.randint(0, p, (B, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32) for _ in range(sets)]
    a, b = a_sets[-1], b_sets[-1]   # the set of the last timed step is the one verified below
    out = torch.zeros((B, L), dtype=torch.int32, device="cuda")
    turn = [0]
    ws_bytes = lib.lib.ntt_workspace_bytes(n, p, B)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        k = turn[0] % sets
        turn[0] += 1
        st = lib.lib.ntt_poly_multiply_batched_device(a_sets[k].data_ptr(), b_sets[k].data_ptr(), out.data_ptr(), n, p, B, n, n, L,
                                                      ws.data_ptr(), ws_bytes, stream)
        lib.check(st)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1:
        # the first collectives create the communicator (seconds): do that before the ramp, so that the barrier in
        # front of the timed region is a short one and the GPUs enter it warm
        dist.barrier()
        tw = torch.zeros(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
    # Clock ramp, untimed and declared in the output line: on MI355X the first ~25 ms of a sustained integer load
    # run up to 40 % slower than the steady state that follows (per-launch trace in profiles/), so a short
    # benchmark would time the transient instead of the throughput.
    if args.ramp_ms > 0:
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    turn[0] = sets - (args.steps % sets)   # so that the last timed step uses the last set
    fence()
    if not args.no_kernel_events:
        lib.lib.ntt_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations from the HIP events recorded on the launch stream during the timed steps
    kernels = {}
    if not args.no_kernel_events:
        names = ctypes.create_string_buffer(512)
        ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int * 16)()
        lib.lib.ntt_profile_read.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        k = lib.lib.ntt_profile_read(names, 512, ms, cnt, 16)
        labels = names.value.decode().split(",") if k > 0 else []
        for i, lab in enumerate(labels):
            kernels[lab] = {"avg_ms": ms[i] / max(cnt[i], 1), "launches": cnt[i]}
        lib.lib.ntt_profile_enable(0)

    # spot-check three products of this rank's batch against the oracle (outside the timed region)
    verified = None
    if rank == 0 and not args.no_verify:
        from oracle_binding import Oracle
        orc = Oracle()
        res = out.cpu().numpy()
        ah, bh = a.cpu().numpy(), b.cpu().numpy()
        verified = all(np.array_equal(res[r, :2 * n - 1].astype(np.uint64), orc.ntt_multiply(ah[r], bh[r], p)) for r in (0, 1, B - 1))

    # what a plain device-to-device copy reaches on this GPU (read + write bytes per second), outside the timed
    # region: the practical ceiling to read the 8 TB/s spec peak against
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); dst = torch.empty_like(src)  # 1 GiB each
        dst.copy_(src); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B / (elapsed / args.steps)
        # Algorithmic bytes of a product under the two-pass model (every transform makes two passes over
        # memory, padding never read, pointwise fused, 2n-1 outputs): 96*n.  Share per kernel: column pass of A
        # and B (read 4n, write 8n each) 24n; fused core (second pass of both forward transforms 32n, pointwise,
        # first pass of the inverse 24n) 56n; inverse column pass (read 8n, write 8n) 16n.  Because the core
        # keeps the spectra on chip, only 24n of its 56n (read 16n, write 8n) is HBM traffic in this design:
        # `moved` prices the kernels on the bytes they must actually move.
        alg = {"k_fwd_strided": 24 * n * B, "k_core_mul": 56 * n * B, "k_inv_strided": 16 * n * B,
               "k_fused8192": 16 * n * B, "k_small": 16 * n * B}
        moved = {"k_fwd_strided": 24 * n * B, "k_core_mul": 24 * n * B, "k_inv_strided": 16 * n * B,
                 "k_fused8192": 16 * n * B, "k_small": 16 * n * B}
        roofline = None
        if kernels and not all(k_ in alg for k_ in kernels):
            # a diagnostic build with other kernels: report their times only
            roofline = {"kernels": {k_: {"avg_ms": v["avg_ms"], "launches": v["launches"]} for k_, v in kernels.items()}}
        elif kernels:
            dom = max(kernels, key=lambda k_: kernels[k_]["avg_ms"])
            dur = kernels[dom]["avg_ms"] * 1e-3
            achieved = alg[dom] / dur / 1e9
            path_gbs = value / world * 96 * n / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(dom, B),
                        "bytes_per_launch": alg[dom], "avg_launch_ms": kernels[dom]["avg_ms"],
                        "bytes_model": "two-pass model, 96*n per product; this kernel's share 56*n (k_core_mul), "
                                       "24*n (k_fwd_strided), 16*n (k_inv_strided)",
                        "moved_bytes_per_launch": moved[dom], "achieved_moved": moved[dom] / dur / 1e9,
                        "frac_moved": moved[dom] / dur / 1e9 / HBM_PEAK_GBS,
                        "kernels": {k_: {"avg_ms": v["avg_ms"], "launches": v["launches"],
                                         "GBps_model": alg[k_] / (v["avg_ms"] * 1e-3) / 1e9,
                                         "GBps_moved": moved[k_] / (v["avg_ms"] * 1e-3) / 1e9} for k_, v in kernels.items()},
                        "measured_copy_GBps": copy_gbs,
                        "path": {"bytes_per_product_model": 96 * n,
                                 "bytes_per_product_moved": sum(moved[k_] for k_ in kernels) // B,
                                 "achieved": path_gbs, "frac": path_gbs / HBM_PEAK_GBS}}
        line = {"metric": "poly-mults/sec at n=2^17 batched", "value": value, "unit": "poly-mults/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": f"batched poly_multiply n={n} p={p}, {B} products per GPU per step, "
                                       f"inputs/outputs resident in HBM (the n = 2^17, p = 998244353 configuration, batched)",
                           "n": n, "p": p, "batch_per_gpu": B, "parallelism": f"batch-sharded x{world}"},
                "ramp_ms_untimed": args.ramp_ms, "input_sets": sets, "verified_vs_oracle": verified, "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, p)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
