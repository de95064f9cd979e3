"""N > 1 path on CPU: two gloo ranks shard a batch (and the CRT primes) the way bench.py / a
multi-GPU caller does, each rank computing its shard with the CPU twin of the library; the
gathered result must equal the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ntt_amd.sharding import crt_prime_owner, crt_unit_range, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntt_amd._binding import NttLibrary
from ntt_amd.sharding import shard_range, crt_prime_owner, crt_unit_range, max_over_ranks, gather_rows
from oracle_binding import Oracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = NttLibrary(os.path.join(ROOT, "tests", "cpu_twin", "libnttamd_cputwin.so"))
orc = Oracle()
# 1. batched products, sharded by row
p, n, batch = 998244353, 2048, 7
rng = np.random.default_rng(123)           # same inputs on every rank
a = rng.integers(0, p, (batch, n), dtype=np.uint32); b = rng.integers(0, p, (batch, n), dtype=np.uint32)
lo, hi = shard_range(batch, rank, world)
local = lib.multiply_batched(a[lo:hi], b[lo:hi], p)
full = gather_rows(local, batch, dist)
ok1 = all(np.array_equal(full[r].astype(np.uint64), orc.ntt_multiply(a[r], b[r], p)) for r in range(batch))
# 2. CRT primes split across ranks, residues exchanged, every rank merges a coefficient slice
P = 1337006139375617
n2 = 1024
A = rng.integers(0, P, n2, dtype=np.uint64); B = rng.integers(0, P, n2, dtype=np.uint64)
primes = [lib.lib.ntt_crt_prime(i) for i in range(4)]
res = np.zeros((4, 2 * n2 - 1), dtype=np.uint64)
for i, q in enumerate(primes):
    if crt_prime_owner(i, world) == rank:
        res[i] = lib.multiply_u64(A % q, B % q, q)
t = torch.from_numpy(res.astype(np.int64))
dist.all_reduce(t)                           # disjoint rows: sum == exchange (cf. MPI_Bcast per prime)
res = t.numpy().astype(np.uint64)
M = 1
for q in primes: M *= q
lo2, hi2 = shard_range(2 * n2 - 1, rank, world)
part = np.zeros(2 * n2 - 1, dtype=np.uint64)
for j in range(lo2, hi2):
    s = 0
    for i, q in enumerate(primes):
        Mi = M // q
        s += Mi * ((int(res[i, j]) * pow(Mi % q, q - 2, q)) % q)
    part[j] = (s % M) % P
t2 = torch.from_numpy(part.astype(np.int64)); dist.all_reduce(t2)   # cf. MPI_Reduce(SUM) of disjoint slices
ok2 = np.array_equal(t2.numpy().astype(np.uint64), orc.crt_ntt_multiply(A, B, P))
# 3. the packaged form of (2): primes over ranks through the per-prime / merge entry points of the C ABI
from ntt_amd.distributed import crt_multiply_distributed
A3 = rng.integers(0, P, (3, 4096), dtype=np.uint64); B3 = rng.integers(0, P, (3, 4096), dtype=np.uint64)
got3 = crt_multiply_distributed(lib, A3, B3, P, dist, torch.device("cpu"))
ok3 = all(np.array_equal(got3[r], orc.crt_ntt_multiply(A3[r], B3[r], P)) for r in range(3))
got4 = crt_multiply_distributed(lib, A[:1000], B[:1000], P, dist, torch.device("cpu"))
ok3 = ok3 and np.array_equal(got4, orc.schoolbook_u64(A[:1000], B[:1000], P))
for bt, nn in ((1, 4), (2, 1), (5, 8), (world, 512), (2 * world + 1, 64)):   # batches below, at and above the rank count
    A5 = rng.integers(0, P, (bt, nn), dtype=np.uint64); B5 = rng.integers(0, P, (bt, nn), dtype=np.uint64)
    got5 = crt_multiply_distributed(lib, A5, B5, P, dist, torch.device("cpu"), all_ranks=False)
    if rank == 0:
        ok3 = ok3 and all(np.array_equal(got5[r], orc.schoolbook_u64(A5[r], B5[r], P)) for r in range(bt))
    else:
        ok3 = ok3 and got5 is None
# the device-resident form: tensors in, this rank's coefficient slice out, buffers reused by a second call
from ntt_amd.distributed import CrtDistributed
ctx = CrtDistributed(lib, P, 512, 3, torch.device("cpu"), dist)
for seed in (1, 2):
    r7 = np.random.default_rng(seed)
    A7 = r7.integers(0, P, (3, 512), dtype=np.uint64); B7 = r7.integers(0, P, (3, 512), dtype=np.uint64)
    sl = ctx.multiply(torch.from_numpy(A7.view(np.int64)), torch.from_numpy(B7.view(np.int64)))
    want7 = np.zeros((3, ctx.Lp), dtype=np.uint64)
    for r in range(3):
        want7[r, :1023] = orc.crt_ntt_multiply(A7[r], B7[r], P)
    ok3 = ok3 and sl.shape == (3, ctx.Lw) and np.array_equal(sl.numpy().view(np.uint64), want7[:, rank * ctx.Lw:(rank + 1) * ctx.Lw])
    ok3 = ok3 and np.array_equal(ctx.gather(sl).numpy().view(np.uint64), want7[:, :1023])
busy = crt_unit_range(2, rank, world)
ok3 = ok3 and (busy[1] > busy[0] or world > 8)          # two rows keep up to eight ranks busy
flag = torch.tensor([1 if ok3 else 0]); dist.all_reduce(flag, op=dist.ReduceOp.MIN); ok3 = bool(flag.item())
# 4. one long product by the four-step transform in the 64-bit field, matrix sharded over the ranks,
#    all-to-all between the two transform axes (3 per product)
from ntt_amd.fourstep import poly_multiply_fourstep
ok4 = True
for n4 in (4096, 1000, 16384, 70000):
    A4 = rng.integers(0, P, n4, dtype=np.uint64); B4 = rng.integers(0, P, n4, dtype=np.uint64)
    got = poly_multiply_fourstep(lib, A4, B4, P, torch.device("cpu"), dist)
    want = orc.crt_ntt_multiply(A4, B4, P) if n4 & (n4 - 1) == 0 else orc.schoolbook_u64(A4, B4, P)
    ok4 = ok4 and np.array_equal(got, want)
# rank 0 only: the product gathered to one rank, the usual way to look at a sharded result
from ntt_amd.fourstep import FourStep
fs = FourStep(lib, P, 5000, torch.device("cpu"), dist, chunks=2)
A6 = rng.integers(0, P, 5000, dtype=np.uint64); B6 = rng.integers(0, P, 5000, dtype=np.uint64)
ok4 = ok4 and all(np.array_equal(fs.column_slab_device(torch.from_numpy(v.view(np.int64)))[:2].numpy(), fs.column_slab(v)[:2].numpy()) for v in (A6, B6))
got6 = fs.gather(fs.multiply_slabs(fs.column_slab_device(torch.from_numpy(A6.view(np.int64))), fs.column_slab(B6)))
ok4 = ok4 and (np.array_equal(got6, orc.schoolbook_u64(A6, B6, P)) if rank == 0 else got6 is None)
flag = torch.tensor([1 if ok4 else 0]); dist.all_reduce(flag, op=dist.ReduceOp.MIN); ok4 = bool(flag.item())   # every rank's verdict
slow = max_over_ranks(float(rank + 1), dist)
if rank == 0:
    print(json.dumps({"batched": bool(ok1), "crt": bool(ok2), "crt_packaged": bool(ok3), "fourstep": bool(ok4), "max": slow, "world": world}))
dist.barrier(); dist.destroy_process_group()
'''


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [crt_prime_owner(i, 2) for i in range(4)] == [0, 1, 0, 1]
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
    # (prime, row) units: one prime per rank at world 4, one prime per pair at world 8 with the rows split inside the pair
    assert [crt_unit_range(1, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert [crt_unit_range(2, r, 8) for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert crt_unit_range(64, 3, 8) == (96, 128)   # rank 3: second half of prime 1's rows


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_ranks_shard_batch_crt_primes_and_fourstep(tmp_path, world):
    """world = 2: two primes per rank; world = 4: one prime per rank and a 4-way all-to-all in the four-step product;
    world = 8: one prime per pair of ranks with the batch split inside the pair."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"batched": True, "crt": True, "crt_packaged": True, "fourstep": True, "max": float(world), "world": world}
