"""ntt_amd: polynomial multiplication modulo NTT-friendly primes on AMD MI355X.

Host-side mirror of the reference programs' multiply interface
(`poly_multiply(a, b, ab, n, p)`, main.cc:92; `NTT_multiply` /
`CRT_NTT_multiply_serial`, main_naive.cc:139,178) over the C ABI of
libnttamd.so.  Everything numeric runs in hand-written HIP kernels; importing
this package on a machine without the built library raises on first use.
"""
from ._binding import NttError, NttLibrary, load_product_library, PRODUCT_LIB  # noqa: F401
from .api import (poly_multiply, poly_multiply_u64, poly_multiply_batched, poly_multiply_batched_u64,  # noqa: F401
                  read_in_file, read_out_file, write_out_file, run_harness)
