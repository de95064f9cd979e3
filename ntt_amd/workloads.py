"""The benchmark configurations beside the headline, as importable functions: synthetic operands generated on the device,
the timed loop of each workload on one rank or across ranks, and the byte models their roofline figures use.  bench.py and
the scripts under tools/ call these; checking the results is the caller's business (it passes a `verify` callable), so
nothing here touches test infrastructure.

Workloads (the reference times every variant of its MPI program from one `mpiexec` run, main_mpi_CRT.cc:556-597):
  small_batched      65536 products of n = 4096, one kernel per workgroup of products (single GPU, or a shard per rank)
  large_modulus      products of n = 2^17 modulo 19 * 2^46 + 1 (the reference's id-4 modulus), a batch per rank
  large_modulus_crt  the same products by the reference's route -- four 31-bit primes and a 128-bit merge -- with the
                     (prime, row) units dealt over the ranks and one all-to-all of coefficient slices (CrtDistributed)
  long_product       one product of n = 2^28 coefficients as a four-step transform, the matrix sharded over the ranks
                     with an all-to-all between the two axes (FourStep)"""
import time

BIG_P = 1337006139375617          # 19 * 2^46 + 1
HBM_PEAK_GBS = 8000.0


def splitmix64_mod(torch, idx, p):
    """splitmix64 of an int64 index tensor (two's complement arithmetic), reduced mod p (< 2^63)."""
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


def fourstep_slab(torch, fs, offset, p=BIG_P, rows_per_pass=1 << 12):
    """Column slab [R][C/G] of the coefficient vector v[i] = splitmix64(i + offset) mod p, i < n (zero beyond), generated on
    this rank's device in row groups so that no full-length temporary exists."""
    slab = torch.empty((fs.R, fs.Cl), dtype=torch.int64, device=fs.device)
    j = torch.arange(fs.rank * fs.Cl, (fs.rank + 1) * fs.Cl, device=fs.device, dtype=torch.int64).view(1, -1)
    for lo in range(0, fs.R, rows_per_pass):
        hi = min(fs.R, lo + rows_per_pass)
        r = torch.arange(lo, hi, device=fs.device, dtype=torch.int64).view(-1, 1)
        i = j + fs.C * r
        v = splitmix64_mod(torch, i + offset, p)
        slab[lo:hi] = torch.where(i < fs.n, v, torch.zeros_like(v))
    return slab


class RankClock:
    """Times a loop of calls the way the headline is timed: device sync + barrier on both sides, every rank's own elapsed
    time kept, the slowest rank's reported as the job's."""

    def __init__(self, torch, dist, device, comm_device):
        self.torch, self.dist, self.device, self.comm_device = torch, dist, device, comm_device
        self.world = dist.get_world_size() if dist is not None else 1

    def sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def fence(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
            self.sync()

    def ramp(self, fn, seconds):
        """Untimed calls for about `seconds`.  Across ranks every rank makes the SAME number of calls -- `fn` may hold a
        collective, and ranks that each went by their own clock would part ways by one call and wait for each other for
        ever: the count comes from the slowest rank's time for one call."""
        t0 = time.perf_counter()
        if self.dist is None:
            while time.perf_counter() - t0 < seconds:
                fn()
                self.sync()
            return
        fn()
        self.sync()
        t = self.torch.tensor([time.perf_counter() - t0], dtype=self.torch.float64, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        calls = min(int(seconds / max(float(t.item()), 1e-6)), 100000)
        for _ in range(calls):
            fn()
            self.sync()

    def time(self, fn, reps, ramp_s=0.15, warm=1):
        """(seconds per call of the slowest rank, [min, max] of the ranks' seconds per call)."""
        for _ in range(warm):
            fn()
        if ramp_s > 0:
            self.ramp(fn, ramp_s)
        self.fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        self.fence()
        dt = (time.perf_counter() - t0) / reps
        if self.dist is None:
            return dt, [dt, dt]
        t = self.torch.tensor([dt, -dt], dtype=self.torch.float64, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        hi, lo = float(t[0].item()), -float(t[1].item())
        return hi, [lo, hi]

    def all_ok(self, ok):
        """A verdict reduced over the ranks: true only when every rank says so."""
        if self.dist is None:
            return bool(ok)
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() == 1.0)


def hbm_roofline(rate_per_gpu, model_bytes, moved_bytes, note=None, issue=None):
    """Roofline block of one configuration: products per second of one GPU x bytes per product against the HBM peak, on the
    two-pass byte model of the path and on the bytes the kernels of this design move; `issue`: the vector-issue figure where
    that is what binds."""
    r = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
         "bytes_per_product_model": model_bytes, "achieved": rate_per_gpu * model_bytes / 1e9,
         "frac": rate_per_gpu * model_bytes / 1e9 / HBM_PEAK_GBS,
         "bytes_per_product_moved": moved_bytes, "achieved_moved": rate_per_gpu * moved_bytes / 1e9,
         "frac_moved": rate_per_gpu * moved_bytes / 1e9 / HBM_PEAK_GBS}
    if note:
        r["note"] = note
    if issue:
        r["issue"] = issue
    return r


# measured butterfly issue ceiling of the chip (tools/ubench_bfly.hip): lazy 32-bit Montgomery butterflies per second
BUTTERFLIES32_PER_S = 4.9e12


# the same for the exact double-precision field of the 64-bit kernels (tools/ubench_bfly64.hip, five waves per SIMD: eleven
# fp64 instructions per butterfly at four cycles each)
BUTTERFLIES_FIELDD_PER_S = 3.37e12


def issue_bound(butterflies_per_product, rate_per_gpu, peak, source="tools/ubench_bfly.hip"):
    return {"butterflies_per_product": butterflies_per_product, "peak_butterflies_per_s": peak,
            "achieved_butterflies_per_s": butterflies_per_product * rate_per_gpu,
            "frac": butterflies_per_product * rate_per_gpu / peak,
            "source": f"butterfly ceiling measured by {source} (static)"}


def small_batched(torch, lib, clock, n=4096, p=998244353, batch=65536, reps=10, verify=None):
    """65536 (per rank) products of n = 4096 in one call: each workgroup holds whole products in LDS."""
    dev = clock.device
    gen = torch.Generator(device=dev); gen.manual_seed(n + batch)
    a = torch.randint(0, p, (batch, n), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, p, (batch, n), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    c = torch.zeros((batch, 2 * n - 1), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    def fn():
        lib.check(lib.lib.ntt_poly_multiply_batched_device(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, p, batch, n, n, 2 * n - 1, None, 0, st))
    dt, spread = clock.time(fn, reps)
    rate = batch / dt
    L = 1 << max(1, (2 * n - 2).bit_length())
    bfly = 3 * (L // 2) * (L.bit_length() - 1)
    entry = {"config": f"{batch} x n={n} per GPU (batched small products)", "n": n, "p": p, "batch_per_gpu": batch, "n_gpus": clock.world,
             "scaling": "weak", "ms_per_call": dt * 1e3, "rank_ms": [s * 1e3 for s in spread], "poly_mults_per_s": rate * clock.world,
             "kernels": lib.describe_plan(n, p),
             "roofline": hbm_roofline(rate, 16 * n, 16 * n, "operands in, 2n-1 coefficients out, 4-byte words; the kernel is bound by vector issue",
                                      issue_bound(bfly, rate, BUTTERFLIES32_PER_S))}
    if verify is not None:
        entry["verified"] = clock.all_ok(verify(a, b, c, n, p))
    return entry


def large_modulus(torch, lib, clock, n=131072, p=BIG_P, batch=64, reps=20, policy="auto", verify=None):
    """`batch` products per rank modulo the large modulus through the general 64-bit entry point; policy 'auto' = the
    direct 64-bit field, 'reference' = four primes + merge on this GPU."""
    dev = clock.device
    gen = torch.Generator(device=dev); gen.manual_seed(n + batch + (clock.dist.get_rank() if clock.dist is not None else 0))
    a = torch.randint(0, p, (batch, n), generator=gen, device=dev, dtype=torch.int64)
    b = torch.randint(0, p, (batch, n), generator=gen, device=dev, dtype=torch.int64)
    c = torch.zeros((batch, 2 * n - 1), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    lib.set_route_policy(policy)
    try:
        def fn():
            lib.check(lib.lib.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, p, batch, n, n, 2 * n - 1, None, 0, st))
        dt, spread = clock.time(fn, reps)
        kernels = lib.describe_plan(n, p)
    finally:
        lib.set_route_policy("auto")
    rate = batch / dt
    L = 1 << max(1, (2 * n - 2).bit_length())
    crt = "k_crt_merge" in kernels or "k_inv_strided_crt" in kernels
    moved = 272 * n if crt else 128 * n
    note = ("per prime 56n (64-bit operand reads 16n -- the residue step rides on the first column pass --, its writes 16n, fused core 24n) "
            "x 4 + inverse column pass of the four primes with the merge in it 48n" if crt else
            "three transforms of one 64-bit array: operand passes 24n + 24n, fused core 48n, result pass 32n; the core is bound by vector issue")
    bfly = 3 * (L // 2) * (L.bit_length() - 1)
    issue = (issue_bound(4 * bfly, rate, BUTTERFLIES32_PER_S) if crt
             else issue_bound(bfly, rate, BUTTERFLIES_FIELDD_PER_S, "tools/ubench_bfly64.hip"))
    route = "four-prime CRT, the reference's route" if crt else "direct 64-bit field, default route"
    entry = {"config": f"large modulus n={n}, {batch} products per GPU ({route})", "n": n, "p": p, "batch_per_gpu": batch,
             "n_gpus": clock.world, "scaling": "weak", "ms_per_call": dt * 1e3, "rank_ms": [s * 1e3 for s in spread],
             "poly_mults_per_s": rate * clock.world, "kernels": kernels,
             "roofline": hbm_roofline(rate, 480 * n if crt else 192 * n, moved, note, issue)}
    if not crt:
        # the byte model of the four-prime route (residue step 48n + 4 x 96n + merge 48n) is the figure this configuration is
        # defined on; the direct route computes the same residues with fewer bytes, so on that model it can exceed the peak
        entry["roofline"]["model_note"] = "192n = 3 transforms x 2 passes x 16 B x 2n of the 64-bit route's own two-pass model"
        entry["roofline"]["on_four_prime_model"] = {"bytes_per_product": 480 * n, "achieved": rate * 480 * n / 1e9,
                                                    "frac": rate * 480 * n / 1e9 / HBM_PEAK_GBS}
    if verify is not None:
        entry["verified"] = clock.all_ok(verify(a, b, c, n, p))
    return entry


def large_modulus_crt_split(torch, lib, clock, n=131072, p=BIG_P, batch=64, reps=20, verify=None):
    """`batch` products IN TOTAL (the same operands on every rank) by the reference's route with the (prime, row) units
    dealt over the ranks: residue products, one all-to-all of coefficient slices, slice merge.  Every rank ends with its
    coefficient slice of every product."""
    from .distributed import CrtDistributed
    dev = clock.device
    gen = torch.Generator(device=dev); gen.manual_seed(n + batch)          # identical on every rank
    a = torch.randint(0, p, (batch, n), generator=gen, device=dev, dtype=torch.int64)
    b = torch.randint(0, p, (batch, n), generator=gen, device=dev, dtype=torch.int64)
    ctx = CrtDistributed(lib, p, n, batch, dev, clock.dist)
    state = {}
    def fn():
        state["out"] = ctx.multiply(a, b)
    dt, spread = clock.time(fn, reps)
    rate = batch / dt                     # whole job: the batch is not replicated
    world = clock.world
    entry = {"config": f"large modulus n={n}, {batch} products in total, (prime, row) units over {world} GPU(s) + all-to-all of coefficient "
                       f"slices + slice merge (the reference's MPI route)", "n": n, "p": p, "batch_total": batch, "n_gpus": world,
             "scaling": "strong", "ms_per_call": dt * 1e3, "rank_ms": [s * 1e3 for s in spread], "poly_mults_per_s": rate,
             "exchange": {"collective": "all_to_all_single", "per_call": 1, "equal_split": ctx.equal_split,
                          "bytes_sent_per_rank": (world - 1) * (ctx.u_hi - ctx.u_lo) * ctx.Lw * 4,
                          "stream_ordered": ctx.stream_ordered},
             "kernels": "per prime: " + lib.describe_plan(n, int(lib.lib.ntt_crt_prime(0))) + "; k_crt_merge",
             "roofline": hbm_roofline(rate / world, 480 * n, 336 * n + 16 * n,
                                      "per GPU: its share of the products x (336n moved by the pipelines and the merge + 16n through the exchange buffers)")}
    if verify is not None:
        entry["verified"] = clock.all_ok(verify(ctx, a, b, state["out"], n, p))
    return entry


def generated_vector(torch, device, n, offset, p=BIG_P, per_pass=1 << 24):
    """v[i] = splitmix64(i + offset) mod p, i < n, on the device (the vector whose column slabs fourstep_slab generates)."""
    v = torch.empty(n, dtype=torch.int64, device=device)
    for lo in range(0, n, per_pass):
        hi = min(n, lo + per_pass)
        v[lo:hi] = splitmix64_mod(torch, torch.arange(lo, hi, device=device, dtype=torch.int64) + offset, p)
    return v


def long_product_direct(torch, lib, clock, log_n=28, p=BIG_P, reps=3, verify=None, warm=2):
    """One product of n = 2^log_n coefficients on ONE GPU through the general 64-bit entry point (the library's own chain of
    passes over the whole array); every rank runs its own replica."""
    dev = clock.device
    n = 1 << log_n
    a, b = generated_vector(torch, dev, n, 0, p), generated_vector(torch, dev, n, 1 << 32, p)
    c = torch.zeros(2 * n - 1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    def fn():
        lib.check(lib.lib.ntt_poly_multiply_batched_device_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, p, 1, n, n, 2 * n - 1, None, 0, st))
    dt, spread = clock.time(fn, reps, ramp_s=0.0, warm=warm)
    L = 1 << max(1, (2 * n - 2).bit_length())
    model = 3 * 2 * 2 * 8 * L
    entry = {"config": f"one product of n=2^{log_n} coefficients, large modulus, one GPU", "n": n, "p": p, "batch": 1, "n_gpus": 1,
             "ms_per_call": dt * 1e3, "rank_ms": [s * 1e3 for s in spread], "poly_mults_per_s": 1.0 / dt,
             "kernels": lib.describe_plan(min(n, 1 << 30), p),
             "roofline": hbm_roofline(1.0 / dt, model, model,
                                      "3 transforms x 2 passes x 16 B x L (the two-pass four-step model); the fused core is bound by vector issue",
                                      issue_bound(3 * (L // 2) * (L.bit_length() - 1), 1.0 / dt, BUTTERFLIES_FIELDD_PER_S, "tools/ubench_bfly64.hip"))}
    if verify is not None:
        entry["verified"] = clock.all_ok(verify(a, b, c, p))
    return entry


def long_product(torch, lib, clock, log_n=28, p=BIG_P, reps=3, chunks=4, verify=None, warm=2):
    """One product of n = 2^log_n coefficients by the four-step transform in its explicit matrix form, the matrix sharded
    over the ranks (FourStep.multiply_slabs); the operands are generated on the devices, slab by slab."""
    from .fourstep import FourStep
    n = 1 << log_n
    fs = FourStep(lib, p, n, clock.device, clock.dist, chunks)
    a, b = fourstep_slab(torch, fs, 0, p), fourstep_slab(torch, fs, 1 << 32, p)
    wa, wb = a.clone(), b.clone()
    state = {}
    # The product overwrites its operand slabs.  The timed calls simply go on multiplying what the slabs then hold: the
    # kernels accept any 64-bit words and their run time does not depend on the values, so the timed region is the
    # product and nothing else (no operand copies); the product that is checked afterwards starts from fresh copies.
    def fn():
        state["c"] = fs.multiply_slabs(wa, wb)
    dt, spread = clock.time(fn, reps, ramp_s=0.0, warm=warm)
    wa.copy_(a); wb.copy_(b)
    fn()
    L = 1 << fs.log_total
    world = clock.world
    per_gpu_model = 3 * 2 * 2 * 8 * L // world          # three transforms x two passes x (read + write) x 8 bytes, this rank's share
    entry = {"config": f"one product of n=2^{log_n} coefficients, four-step matrix {fs.R}x{fs.C} over {world} GPU(s)", "n": n, "p": p,
             "n_gpus": world, "scaling": "strong", "ms_per_call": dt * 1e3,
             "rank_ms": [s * 1e3 for s in spread], "poly_mults_per_s": 1.0 / dt,
             "exchange": {"collective": "batched point-to-point (all-to-all)", "per_product": 3 if world > 1 else 0, "pieces": fs.chunks,
                          "bytes_sent_per_rank_per_exchange": (world - 1) * fs.Rl * fs.Cl * 8, "stream_ordered": fs.stream_ordered},
             "roofline": hbm_roofline(1.0 / dt, per_gpu_model, per_gpu_model,
                                      "per GPU: its share of 3 transforms x 2 passes x 16 B x L (the two-pass four-step model); the fused core is "
                                      "bound by vector issue",
                                      issue_bound(3 * (L // 2) * (L.bit_length() - 1) // world, 1.0 / dt, BUTTERFLIES_FIELDD_PER_S,
                                                  "tools/ubench_bfly64.hip"))}
    if verify is not None:
        entry["verified"] = clock.all_ok(verify(fs, a, b, state["c"], p))
    return entry
