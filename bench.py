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
import faulthandler
import glob
import json
import os
import sys
import threading
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
    """(HBM bytes per launch of `kernel`, source) from the newest committed PMC summary (profiles/traffic_*.json), which
    was collected at the batch size recorded in it (256 when absent); traffic is proportional to the batch.  The figure
    is static: it describes the kernels as they were when that profile was taken, not this run."""
    def newest_headline_first(path):      # traffic_<tag>.json (the headline run of a tag) before traffic_<tag>_<workload>.json
        tag, _, workload = os.path.basename(path)[len("traffic_"):-len(".json")].partition("_")
        return (tag, workload == "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")), key=newest_headline_first, reverse=True):   # newest summary that has this kernel
        try:
            with open(path) as f:
                d = json.load(f)
            per_launch = d.get(kernel, {}).get("hbm_bytes_per_launch")
            if per_launch is not None and d.get("_meta", {}).get("n", 131072) == 131072:
                return per_launch * batch / d.get("_meta", {}).get("batch", 256), f"profiles/{os.path.basename(path)} (static, rocprofv3 --pmc)"
        except Exception:
            continue
    return None, None


def valu_roofline(kernel, batch, avg_ms):
    """Integer-issue roofline of one kernel: vector instructions per launch (SQ_INSTS_VALU of the newest committed
    profiles/*_pmc.json, scaled to the batch) against the rate at which the chip can issue that instruction mix
    (profiles/valu_issue.json: cycles per wave-instruction per SIMD measured by tools/ubench_valu.hip for plain and
    multiply-class instructions, and the kernel's share of multiply-class instructions from its disassembly)."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_issue.json")) as f:
            iss = json.load(f)
        pmc, used = None, None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):   # newest profile that has this kernel
            with open(path) as f:
                cand = json.load(f)
            if kernel in cand and "SQ_INSTS_VALU" in cand[kernel]:
                pmc, used = cand, path
                break
        insts = pmc[kernel]["SQ_INSTS_VALU"] * batch / iss.get("pmc_batch", 256)
        mix = iss["kernels"][kernel]["multiply_class_fraction"]
        cyc = mix * iss["cycles_per_wave_inst"]["multiply_class"] + (1 - mix) * iss["cycles_per_wave_inst"]["plain"]
        peak = iss["simds"] * iss["clock_GHz"] * 1e9 / cyc
        achieved = insts / (avg_ms * 1e-3)
        return {"kernel": kernel, "wave_insts_per_launch": insts, "multiply_class_fraction": mix,
                "issue_cycles_per_wave_inst": cyc, "peak_wave_insts_per_s": peak, "achieved_wave_insts_per_s": achieved,
                "frac": achieved / peak,
                "source": f"profiles/{os.path.basename(used)} + profiles/valu_issue.json (static)"}
    except Exception:
        return None


def other_configurations(torch, dist, lib, clock, sizes):
    """The remaining configurations of the baseline, timed after the headline by the same rank processes (operands
    generated on the devices, steady state after an untimed ramp) and reported beside it, never as `value`.
      one rank:      config 3 (65536 x n = 4096), config 4 by both routes, config 5 directly and in its explicit four-step form
      several ranks: config 3 and config 4 sharded by batch (weak scaling), config 4 by the reference's MPI route -- (prime, row)
                     units over the ranks, one all-to-all, slice merge -- and config 5 with the matrix sharded over the ranks
                     (strong scaling), the way one `mpiexec` run of the reference times all its variants (main_mpi_CRT.cc:556-597).
    Every entry carries a roofline block and a verification verdict reduced over the ranks.  The checks are test
    infrastructure (tests/evalcheck.py: evaluation identities in torch integer arithmetic); bit-exactness against the oracle
    is the business of the GPU test suite."""
    from ntt_amd import workloads as W
    import evalcheck
    world, rank = clock.world, (dist.get_rank() if clock.dist is not None else 0)
    g = torch.Generator(device="cpu"); g.manual_seed(0xE7A1)        # the same evaluation points on every rank

    def point(p):
        return int(torch.randint(2, min(p, 1 << 62) - 1, (1,), generator=g).item())

    def rows_identity(a, b, c, n, p, sample=256):
        """A_r(x) B_r(x) = C_r(x) mod p on a sample of rows (all of them when there are few) at one random point."""
        x = point(p)
        rows = a.shape[0]
        pick = torch.arange(0, rows, max(1, rows // sample), device=a.device)
        def vals(t):
            t = t[pick].to(torch.int64)
            if t.dtype != a.dtype or a.dtype == torch.int32:
                t = t & 0xFFFFFFFF
            return evalcheck.evaluate_rows(t, x, p)
        va, vb, vc = vals(a), vals(b), vals(c)
        return all(x_ * y_ % p == z_ for x_, y_, z_ in zip(va, vb, vc))

    def crt_slice_equals_direct(ctx, a, b, out, n, p):
        """This rank's coefficient slice from the split CRT route against the same slice of the products computed here by the
        default route (the two routes are bit-identical wherever both apply)."""
        ref = torch.zeros((ctx.batch, ctx.Lp), dtype=torch.int64, device=a.device)
        st = torch.cuda.current_stream(a.device).cuda_stream if a.device.type == "cuda" else None
        lib.check(lib.lib.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), ref.data_ptr(), n, p, ctx.batch, n, n, ctx.Lp, None, 0, st))
        return bool(torch.equal(out, ref[:, ctx.rank * ctx.Lw:(ctx.rank + 1) * ctx.Lw]))

    def long_identity(a, b, c, p):
        ok = True
        for _ in range(2):
            x = point(p)
            ok = ok and evalcheck.evaluate(a, x, p) * evalcheck.evaluate(b, x, p) % p == evalcheck.evaluate(c, x, p)
        return ok

    def slabs_identity(fs, a, b, c, p):
        """A(x) B(x) = C(x) at two random points; every rank evaluates its column slabs, the shares are summed over the ranks."""
        ok = True
        rows_in = min(fs.R, -(-fs.n // fs.C))
        for _ in range(2):
            x = point(p)
            share = [evalcheck.evaluate_slab(t, fs.rank * fs.Cl, fs.C, x, p, rows) for t, rows in ((a, rows_in), (b, rows_in), (c, fs.R))]
            if clock.dist is not None:
                parts = [None] * world
                dist.all_gather_object(parts, share)
                share = [sum(q[i] for q in parts) % p for i in range(3)]
            ok = ok and share[0] * share[1] % p == share[2]
        return ok

    c3, c4, c5 = sizes["c3"], sizes["c4"], sizes["c5"]
    if world == 1:
        jobs = [lambda: W.small_batched(torch, lib, clock, c3["n"], 998244353, c3["batch"], 10, rows_identity),
                lambda: W.large_modulus(torch, lib, clock, c4["n"], W.BIG_P, c4["batch"], 20, "auto", rows_identity),
                lambda: W.large_modulus(torch, lib, clock, c4["n"], W.BIG_P, c4["batch"], 20, "reference", rows_identity),
                lambda: W.long_product_direct(torch, lib, clock, c5["log_n"], W.BIG_P, 3, long_identity),
                lambda: W.long_product(torch, lib, clock, c5["log_n"], W.BIG_P, 3, 4, slabs_identity)]
    else:
        jobs = [lambda: W.small_batched(torch, lib, clock, c3["n"], 998244353, c3["batch"], 10, rows_identity),
                lambda: W.large_modulus(torch, lib, clock, c4["n"], W.BIG_P, c4["batch"], 20, "auto", rows_identity),
                lambda: W.large_modulus_crt_split(torch, lib, clock, c4["n"], W.BIG_P, c4["batch"], 20, crt_slice_equals_direct),
                lambda: W.long_product(torch, lib, clock, c5["log_n"], W.BIG_P, 3, 4, slabs_identity)]
    out = []
    for job in jobs:
        if world == 1:
            try:
                out.append(job())
            except Exception as e:   # a failure here must not take the headline line with it
                out.append({"error": f"{type(e).__name__}: {e}"})
        else:
            out.append(job())        # across ranks a failure must end the run: a rank that skipped a workload would leave the others waiting
        if clock.device.type == "cuda":
            torch.cuda.empty_cache()
    return out if rank == 0 else None


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def launch_ranks(args, argv):
    """`bench.py --gpus N` started as one plain process: fan out to N rank processes (one per GPU), the way the
    reference's MPI programs are started with `mpiexec -np N` (qsub_mpi.sh:16), and relay rank 0's line.  This parent
    never touches torch or the GPU; the ranks are fresh children that find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, exactly as under torch.distributed.run."""
    import subprocess
    port = str(free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # rank 0's output is read by a thread while every rank is watched: when one rank ends badly the others (blocked in a
    # collective that can no longer complete) are ended too, by their own process ids
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, q in enumerate(procs):
            if codes[i] is None:
                codes[i] = q.poll()
        if any(c not in (None, 0) for c in codes):
            for i, q in enumerate(procs):
                if codes[i] is None:
                    q.terminate()
            for i, q in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = q.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        q.kill()
                        codes[i] = q.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out0 = captured[0] if captured else ""
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    for l in out0.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if any(codes) or not lines:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed (exit codes {codes})")
    line = json.loads(lines[-1])
    if line.get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} GPUs but the ranks report n_gpus={line.get('n_gpus')}")
    print(lines[-1], flush=True)


def verify_products(torch, a, b, out, n, p, rows, oracle_rows):
    """Every product of the batch is checked on the device through the evaluation identity A(x) * B(x) = C(x) (mod p) at
    two seeded random points (a wrong coefficient survives one point with probability < 2n/p), and `oracle_rows` of them
    are compared bit for bit with the CPU oracle.  Runs after the timed region; test infrastructure, not product code."""
    import numpy as np
    from oracle_binding import Oracle
    dev = a.device
    g = torch.Generator(device="cpu"); g.manual_seed(0xC0DE)
    ok_eval = True
    for _ in range(2):
        x = int(torch.randint(2, p - 1, (1,), generator=g).item())
        pw = torch.ones(1, dtype=torch.int64, device=dev)
        step = x
        while pw.numel() < 2 * n:       # powers x^0 .. x^(2n-1) by doubling
            pw = torch.cat([pw, pw * step % p])
            step = step * step % p
        def ev(t, count):
            acc = torch.zeros(t.shape[0], dtype=torch.int64, device=dev)
            for lo in range(0, t.shape[0], 32):     # 32 rows at a time keeps the 64-bit temporaries small
                v = t[lo:lo + 32, :count].to(torch.int64) & 0xFFFFFFFF
                acc[lo:lo + 32] = (v * pw[:count] % p).sum(dim=1) % p
            return acc
        lhs = ev(a[:rows], n) * ev(b[:rows], n) % p
        ok_eval = ok_eval and bool(torch.equal(lhs, ev(out[:rows], 2 * n - 1)))
    orc = Oracle()
    pick = sorted(set([0, 1, rows - 1] + list(range(0, rows, max(1, rows // max(1, oracle_rows))))))[:max(3, oracle_rows)]
    pick = [r for r in pick if r < rows]
    ah = a[pick].cpu().numpy().view(np.uint32); bh = b[pick].cpu().numpy().view(np.uint32); oh = out[pick].cpu().numpy().view(np.uint32)
    ok_exact = all(np.array_equal(oh[i, :2 * n - 1].astype(np.uint64), orc.ntt_multiply(ah[i], bh[i], p)) for i in range(len(pick)))
    return {"all_products_evaluation_identity": ok_eval, "rows_checked": rows, "oracle_rows_bit_exact": len(pick),
            "oracle_ok": ok_exact, "ok": bool(ok_eval and ok_exact)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed run of the same workload before the warm-up steps, long enough for the GPU's clock "
                         "management to settle (it dips for the first ~25 ms of a sustained load); 0 disables")
    ap.add_argument("--n", "--size", dest="n", type=int, default=131072,
                    help="coefficients per operand (--size is the spelling to use under torch.distributed.run, whose own parser claims --n)")
    ap.add_argument("--p", type=int, default=998244353)
    ap.add_argument("--batch", type=int, default=256, help="products per GPU per step")
    ap.add_argument("--input-sets", type=int, default=4,
                    help="distinct operand sets the steps rotate through, so that no step finds its inputs in the 256 MB "
                         "Infinity Cache from the step before")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the timing of the other configurations after the headline")
    ap.add_argument("--other-sizes", default=None,
                    help="sizes of the other configurations as n3,batch3,n4,batch4,log_n5 (default 4096,65536,131072,64,28; the CPU "
                         "rehearsal uses 256,8,1024,4,12)")
    ap.add_argument("--others-deadline-s", type=float, default=240.0,
                    help="with several ranks: seconds the other configurations may take before every rank gives them up and "
                         "rank 0 prints the headline line without them")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of the library instead of the shipped one")
    ap.add_argument("--no-verify", action="store_true", help="diagnostic: skip the result checks (ablated builds)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="rehearsal of the multi-rank control flow without GPUs: CPU tensors, gloo, and the library given by "
                         "--lib (the test-only CPU twin); the line is marked as a rehearsal and is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])

    import numpy as np
    import torch
    import torch.distributed as dist
    from ntt_amd._binding import load_product_library

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    cpu = args.rehearse_cpu
    if cpu and not args.lib:
        raise SystemExit("--rehearse-cpu needs --lib (the CPU twin); the shipped library has no CPU path")
    backend = None
    if not cpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU")
        backend = os.environ.get("NTT_BENCH_BACKEND", "nccl")
        if world > torch.cuda.device_count() and backend == "nccl":
            raise SystemExit(f"bench.py: --gpus {world} needs {world} GPUs, this node shows {torch.cuda.device_count()}")
        device_index = local_rank % torch.cuda.device_count()  # one rank per GPU; wraps only in gloo rehearsals
        torch.cuda.set_device(device_index)
        dev = torch.device("cuda", device_index)
    else:
        backend, dev = "gloo", torch.device("cpu")
    if world > 1:
        # RCCL carries the barrier and the max-reduce of the clock (the products themselves need no collective);
        # NTT_BENCH_BACKEND=gloo rehearses the multi-rank logic on a box with fewer GPUs than ranks
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
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

    def sync():
        if not cpu:
            torch.cuda.synchronize()

    gen = torch.Generator(device=dev); gen.manual_seed(0x5EED0000 + rank)
    sets = max(1, args.input_sets)
    a_sets = [torch.randint(0, p, (B, n), generator=gen, device=dev, dtype=torch.int64).to(torch.int32) for _ in range(sets)]
    b_sets = [torch.randint(0, p, (B, n), generator=gen, device=dev, dtype=torch.int64).to(torch.int32) for _ in range(sets)]
    a, b = a_sets[-1], b_sets[-1]   # the set of the last timed step is the one verified below
    out = torch.zeros((B, L), dtype=torch.int32, device=dev)
    turn = [0]
    ws_bytes = lib.lib.ntt_workspace_bytes(n, p, B)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    stream = None if cpu else torch.cuda.current_stream().cuda_stream

    def step():
        k = turn[0] % sets
        turn[0] += 1
        st = lib.lib.ntt_poly_multiply_batched_device(a_sets[k].data_ptr(), b_sets[k].data_ptr(), out.data_ptr(), n, p, B, n, n, L,
                                                      ws.data_ptr(), ws_bytes, stream)
        lib.check(st)

    def fence():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    collective_info = None
    if world > 1:
        # the first collectives create the communicator (seconds): do that before the ramp, so that the barrier in
        # front of the timed region is a short one and the GPUs enter it warm
        dist.barrier()
        tw = torch.zeros(1, dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        sync()
        # attestation that the ranks really talk to one another: every rank contributes a one, the sum must be the world size
        ones = torch.ones(1, dtype=torch.float64, device=comm_dev)
        dist.all_reduce(ones)
        sync()
        collective_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "all_reduce_of_ones": float(ones.item()),
                           "devices": torch.cuda.device_count() if not cpu else 0}
    # Clock ramp, untimed and declared in the output line: on MI355X the first ~25 ms of a sustained integer load
    # run up to 40 % slower than the steady state that follows (per-launch trace in profiles/), so a short
    # benchmark would time the transient instead of the throughput.
    if args.ramp_ms > 0:
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(8):
                step()
            sync()
    for _ in range(args.warmup):
        step()
    turn[0] = sets - (args.steps % sets)   # so that the last timed step uses the last set
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Every product of this rank's last step is checked (evaluation identity on the device) and 1/16 of them bit for
    # bit against the oracle; the verdicts of all ranks are combined.  Outside the timed region.
    verified = None
    if not args.no_verify:
        verified = verify_products(torch, a, b, out, n, p, B, max(1, B // 16))
        if world > 1:
            flag = torch.tensor([1.0 if verified["ok"] else 0.0], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verified["all_ranks_ok"] = bool(flag.item() == 1.0)
            verified["ok"] = verified["all_ranks_ok"]

    # Per-kernel durations: a separate, untimed pass of the same steps with every launch bracketed by HIP events on the
    # launch stream (the headline loop above runs without them).
    kernels = {}
    if not args.no_kernel_events and rank == 0:
        # (the checks above left the GPU idle: ramp and warm up again, or the pass would time the clock transient)
        t_ramp = time.perf_counter()
        while args.ramp_ms > 0 and (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(8):
                step()
            sync()
        for _ in range(args.warmup):
            step()
        lib.lib.ntt_profile_enable(1)
        for _ in range(min(args.steps, 100)):
            step()
        sync()
        names = ctypes.create_string_buffer(512)
        ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int * 16)()
        lib.lib.ntt_profile_read.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        k = lib.lib.ntt_profile_read(names, 512, ms, cnt, 16)
        labels = names.value.decode().split(",") if k > 0 else []
        for i, lab in enumerate(labels):
            kernels[lab] = {"avg_ms": ms[i] / max(cnt[i], 1), "launches": cnt[i]}
        lib.lib.ntt_profile_enable(0)

    # what a plain device-to-device copy reaches on this GPU (read + write bytes per second), outside the timed
    # region: the practical ceiling to read the 8 TB/s spec peak against
    copy_gbs = None
    if rank == 0 and not cpu:
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
        # Algorithmic bytes of a product under the two-pass model (every transform makes two passes over memory,
        # padding never read, pointwise fused, 2n-1 outputs): 96*n.  `roofline.achieved` prices the whole path of one
        # GPU on that model -- products per second x 96n -- which is what the HBM target of the path refers to; the
        # kernels of this design move 64*n per product (the spectra never leave the chip), `achieved_moved`.
        # Per kernel: share of the model (column pass of A and B 24n; fused core 56n; inverse column pass 16n) and
        # bytes it must really move (24n / 24n / 16n).
        alg = {"k_fwd_strided": 24 * n * B, "k_core_mul": 56 * n * B, "k_inv_strided": 16 * n * B,
               "k_fused8192": 16 * n * B, "k_small": 16 * n * B}
        moved = {"k_fwd_strided": 24 * n * B, "k_core_mul": 24 * n * B, "k_inv_strided": 16 * n * B,
                 "k_fused8192": 16 * n * B, "k_small": 16 * n * B}
        per_gpu = value / world
        multipass = n > 4096
        model_bytes = 96 * n if multipass else 16 * n
        moved_bytes = 64 * n if multipass else 16 * n
        path_gbs, moved_gbs = per_gpu * model_bytes / 1e9, per_gpu * moved_bytes / 1e9
        roofline = {"bound": "hbm", "scope": "whole path of one GPU: products/s x algorithmic bytes per product",
                    "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": path_gbs / HBM_PEAK_GBS,
                    "bytes_per_product_model": model_bytes, "bytes_per_product_moved": moved_bytes,
                    "achieved_moved": moved_gbs, "frac_moved": moved_gbs / HBM_PEAK_GBS,
                    "traffic": None, "traffic_source": None, "measured_copy_GBps": copy_gbs}
        if kernels and not all(v["avg_ms"] > 0 for v in kernels.values()):
            roofline["kernels"] = {k_: {"avg_ms": v["avg_ms"], "launches": v["launches"]} for k_, v in kernels.items()}
        elif kernels and all(k_ in alg for k_ in kernels):
            dom = max(kernels, key=lambda k_: kernels[k_]["avg_ms"])
            dur = kernels[dom]["avg_ms"] * 1e-3
            traffic, traffic_src = load_traffic(dom, B)
            roofline.update({
                "traffic": traffic, "traffic_source": traffic_src, "traffic_kernel": dom,
                "dominant_kernel": {"kernel": dom, "avg_launch_ms": kernels[dom]["avg_ms"],
                                    "bytes_per_launch_model": alg[dom], "GBps_model": alg[dom] / dur / 1e9,
                                    "frac_model": alg[dom] / dur / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_launch_moved": moved[dom], "GBps_moved": moved[dom] / dur / 1e9,
                                    "frac_moved": moved[dom] / dur / 1e9 / HBM_PEAK_GBS},
                "kernels": {k_: {"avg_ms": v["avg_ms"], "launches": v["launches"],
                                 "GBps_model": alg[k_] / (v["avg_ms"] * 1e-3) / 1e9,
                                 "GBps_moved": moved[k_] / (v["avg_ms"] * 1e-3) / 1e9} for k_, v in kernels.items()},
                "kernel_times": "HIP events on the launch stream, separate untimed pass after the timed steps",
                "valu": valu_roofline(dom, B, kernels[dom]["avg_ms"])})
        elif kernels:   # a diagnostic build with other kernels: report their times only
            roofline["kernels"] = {k_: {"avg_ms": v["avg_ms"], "launches": v["launches"]} for k_, v in kernels.items()}
        line = {"metric": "poly-mults/sec at n=2^17 batched", "value": value, "unit": "poly-mults/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": f"batched poly_multiply n={n} p={p}, {B} products per GPU per step, "
                                       f"inputs/outputs resident in HBM (the n = 2^17, p = 998244353 configuration, batched)",
                           "n": n, "p": p, "batch_per_gpu": B, "parallelism": f"batch-sharded x{world}"},
                "ramp_ms_untimed": args.ramp_ms, "input_sets": sets,
                "verified_vs_oracle": None if verified is None else verified["ok"], "verification": verified,
                "roofline": roofline}
        if cpu:
            line["rehearsal"] = "CPU twin over gloo: control-flow rehearsal, not a measurement"
        if collective_info is not None:
            line["rccl"] = collective_info

    # The other configurations, timed by the same rank processes (collective across the ranks: every rank takes part).
    # The headline line is complete by now.  Across ranks these workloads exchange data, and a rank that fails or stalls
    # in one of them would leave the others waiting in a collective for ever: every rank therefore arms a deadline, and
    # when it passes (or this rank's workloads raise) rank 0 prints the headline line with a note in place of the other
    # configurations and the process ends at once, without waiting for the collective it is stuck in.
    others = None
    run_others = not args.no_other_configs and (cpu or not args.lib)
    if run_others:
        from ntt_amd.workloads import RankClock
        del a_sets, b_sets, ws
        if not cpu:
            torch.cuda.empty_cache()
        if args.other_sizes:
            v = [int(x) for x in args.other_sizes.split(",")]
        else:
            v = [256, 8, 1024, 4, 12] if cpu else [4096, 65536, 131072, 64, 28]
        sizes = {"c3": {"n": v[0], "batch": v[1]}, "c4": {"n": v[2], "batch": v[3]}, "c5": {"log_n": v[4]}}

        once = threading.Lock()      # held for good by whoever prints the line

        def abandon(reason):
            if not once.acquire(blocking=False):
                return
            if rank == 0:
                line["other_configs"] = [{"error": reason}]
                print(json.dumps(line), flush=True)
            sys.stderr.write(f"bench.py rank {rank}: other configurations abandoned: {reason}\n")
            sys.stderr.flush()
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)      # where this rank was
            os._exit(0)

        deadline = None
        if world > 1:
            dist.barrier()
            # rank 0 first, so that its line is out before any other rank's exit is noticed
            deadline = threading.Timer(args.others_deadline_s + (0 if rank == 0 else 5),
                                       abandon, [f"not finished within {args.others_deadline_s:.0f} s"])
            deadline.daemon = True
            deadline.start()
        clock = RankClock(torch, dist if world > 1 else None, dev, comm_dev)
        try:
            others = other_configurations(torch, dist, lib, clock, sizes)
        except Exception as e:      # world > 1 only (a single rank records the error per workload and goes on)
            if rank == 0:
                abandon(f"{type(e).__name__}: {e}")
            sys.stderr.write(f"bench.py rank {rank}: {type(e).__name__}: {e}\n")
            sys.stderr.flush()
            while True:             # the other ranks are waiting for this one: the deadline ends all of them
                time.sleep(1)
        if deadline is not None:
            deadline.cancel()
            if not once.acquire(blocking=False):     # the deadline fired while the last workload returned: it ends the process
                while True:
                    time.sleep(1)

    if rank == 0:
        if others is not None:
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline and not cpu:
            line["cpu_baseline"] = cpu_baseline(n, p)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
