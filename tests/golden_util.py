"""Iterates the committed golden vectors (tests/golden): reference data files (.in/.out text),
their n = 131072 cases re-encoded as .npz, and outputs of the compiled reference (ref_*.npz)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_in_file(path):
    """Independent pure-Python reader of the `.in` format (n p, then a, then b), also the check on the library's scanner."""
    with open(path, "rb") as f:
        tok = f.read().split()
    n, p = int(tok[0]), int(tok[1])
    vals = np.array(tok[2:2 + 2 * n], dtype=np.uint64)
    assert vals.size == 2 * n, path
    return n, p, vals[:n].copy(), vals[n:].copy()


def read_out_file(path, n):
    with open(path, "rb") as f:
        tok = f.read().split()
    vals = np.array(tok[:2 * n - 1], dtype=np.uint64)
    assert vals.size == 2 * n - 1, path
    return vals


def iter_golden(max_n=None):
    for f in sorted(os.listdir(GOLDEN)):
        path = os.path.join(GOLDEN, f)
        if f.endswith(".in"):
            n, p, a, b = read_in_file(path)
            out = read_out_file(path[:-3] + ".out", n)
        elif f.endswith(".npz"):
            z = np.load(path)
            n, p = int(z["n"]), int(z["p"])
            a, b, out = z["a"].astype(np.uint64), z["b"].astype(np.uint64), z["out"].astype(np.uint64)
        else:
            continue
        if max_n is None or n <= max_n:
            yield f, n, p, a, b, out
