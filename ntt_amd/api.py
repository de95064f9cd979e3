"""Python face of the drop-in boundary plus the reference's text harness.

Function names and argument meaning follow the reference: `poly_multiply`
(main.cc:92-98), the 64-bit dispatch of main_naive.cc:253-257, and the
`fRead` / `fCheck` / `fWrite` file conventions (main_naive.cc:13-67).
"""
import os
import time

import numpy as np

from ._binding import load_product_library


def poly_multiply(a, b, n, p):
    """ab = a*b mod p for int32 coefficient arrays of length n (main.cc:92 signature)."""
    a = np.asarray(a, dtype=np.int32)[:n]; b = np.asarray(b, dtype=np.int32)[:n]
    return load_product_library().multiply_i32(a, b, p)


def poly_multiply_u64(a, b, n, p):
    """64-bit entry point, the slot of the dispatch of main_naive.cc:253-257: the 32-bit NTT when p fits; for larger p the
    residues the reference's four-prime CRT gives, computed in the field of p itself when p is an NTT-friendly prime
    (the reference's 1337006139375617 is) and through the four primes otherwise -- see set_route_policy."""
    a = np.asarray(a, dtype=np.uint64)[:n]; b = np.asarray(b, dtype=np.uint64)[:n]
    return load_product_library().multiply_u64(a, b, p)


def poly_multiply_batched(a, b, p, n_gpus=0):
    """Rows of a and b are independent products; the batch is sharded over n_gpus devices."""
    return load_product_library().multiply_batched(a, b, p, n_gpus)


def poly_multiply_batched_u64(a, b, p, n_gpus=0):
    return load_product_library().multiply_batched_u64(a, b, p, n_gpus)


def set_route_policy(policy):
    """'auto' (default: the fastest exact route) or 'reference' (the four-prime CRT wherever the reference uses it, i.e.
    for every p > 2^32); returns the previous policy.  Results are identical either way."""
    return load_product_library().set_route_policy(policy)


# ---- text harness (main_naive.cc:13-67) ------------------------------------
# The parsing and formatting run inside the library (mapped file, one scan); these are host-only
# calls and work without a GPU.
def read_in_file(path, sidecar=False):
    """`n p` then n coefficients of a and n of b, whitespace separated -> (n, p, a, b)."""
    return load_product_library().read_in_file(path, sidecar)


def read_out_file(path, n):
    return load_product_library().read_out_file(path, 2 * n - 1)


def check_out_file(path, ab):
    return load_product_library().check_out_file(path, ab)


def write_out_file(path, ab):
    """One value per line, as fWrite does."""
    load_product_library().write_out_file(path, ab)


def run_harness(data_dir, out_dir=None, ids=range(0, 5), stream=None):
    """Loop of the reference main (main_naive.cc:243-268): read <id>.in, multiply, compare
    with <id>.out, print the same two lines, optionally write <out_dir>/<id>.out.
    Missing inputs are skipped (the reference reads uninitialised sizes instead).
    Returns {id: bool correct}."""
    import sys
    stream = stream or sys.stdout
    results = {}
    for i in ids:
        fin = os.path.join(data_dir, f"{i}.in")
        if not os.path.exists(fin):
            continue
        n, p, a, b = read_in_file(fin)
        t0 = time.perf_counter()
        ab = poly_multiply_u64(a, b, n, p)
        ms = (time.perf_counter() - t0) * 1e3
        ok = None
        fout = os.path.join(data_dir, f"{i}.out")
        if os.path.exists(fout):
            ok = check_out_file(fout, ab)
            print("多项式乘法结果正确" if ok else "多项式乘法结果错误", file=stream)
        print(f"average latency for n = {n} p = {p} : {ms:g} (us) ", file=stream)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
            write_out_file(os.path.join(out_dir, f"{i}.out"), ab)
        results[i] = ok
    return results
