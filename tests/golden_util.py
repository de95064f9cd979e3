"""Iterates the committed golden vectors (tests/golden): reference data files (.in/.out text),
their n = 131072 cases re-encoded as .npz, and outputs of the compiled reference (ref_*.npz)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def iter_golden(max_n=None):
    from ntt_amd.api import read_in_file, read_out_file
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
