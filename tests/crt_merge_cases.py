"""The CRT recombination on its own (ntt_crt_merge_device) against Python integers: random residues, the corners of the
residue ranges and values x next to 0 and to M - 1, for final moduli on both sides of every switch in the kernel
(double-precision tail up to 1.5e15, integer tail above; even, tiny and 64-bit moduli).  Shared by the CPU-twin and the
GPU test; `alloc(np_array)` gives (pointer, keep-alive, read-back) for the library under test."""
import numpy as np

PRIMES = [1004535809, 1224736769, 469762049, 998244353]
M = 1
for q in PRIMES:
    M *= q
MODULI = [2, 3, 97, 1000000007, 2**32 - 5, 2**32 + 15, 1337006139375617, 1500000000000000, 1500000000000001, 2**52 + 21,
          2**60 - 93, 2**63 + 29, 2**64 - 59, 2**64 - 1]


def residues_of(xs):
    return np.array([[x % q for x in xs] for q in PRIMES], dtype=np.uint32)


def cases(rng, count=3000):
    xs = [0, 1, 2, M - 1, M - 2, M // 2, M // 2 + 1]
    xs += [q - 1 for q in PRIMES] + [q for q in PRIMES] + [M - q for q in PRIMES] + [M // q for q in PRIMES] + [M // q - 1 for q in PRIMES]
    xs += [int.from_bytes(rng.bytes(15), "little") % M for _ in range(count)]
    return xs


def check(lib, to_device, from_device, p, xs):
    res = residues_of(xs)                     # [4][count]
    count = res.shape[1]
    d_res = to_device(res)
    d_out = to_device(np.zeros(count, dtype=np.uint64))
    lib.check(lib.lib.ntt_crt_merge_device(d_res["ptr"], count, d_out["ptr"], count, p, None))
    got = from_device(d_out)
    want = np.array([x % p for x in xs], dtype=np.uint64)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (p, [(xs[i], int(got[i]), int(want[i])) for i in bad[:3]])
