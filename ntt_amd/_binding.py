"""ctypes binding of the libnttamd C ABI (include/ntt_amd.h).

The binding class only declares signatures and converts arguments; all
arithmetic happens inside the shared library.  `load_product_library()` loads
the HIP build that ships in this package and raises if it is missing -- there
is no software fallback.
"""
import ctypes
import os

import numpy as np

NTT_OK = 0
_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libnttamd.so")

c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_i32p = ctypes.POINTER(ctypes.c_int32)

EXPORTS = [
    "ntt_init", "ntt_shutdown", "ntt_device_count", "ntt_last_error", "ntt_set_route_policy", "ntt_release_workspace",
    "poly_multiply", "poly_multiply_u64", "ntt_poly_multiply_i32", "ntt_poly_multiply_u64",
    "ntt_poly_multiply_batched", "ntt_poly_multiply_batched_u64",
    "ntt_workspace_bytes", "ntt_poly_multiply_batched_device", "ntt_poly_multiply_batched_device_u64",
    "ntt_poly_multiply_batched_device_noalloc",
    "ntt_crt_merge_device", "ntt_crt_residue_product_device", "ntt_crt_prime", "ntt_describe_plan",
    "ntt64_transform_device", "ntt64_fourstep_rows_device", "ntt64_fourstep_cols_device",
    "ntt64_fourstep_rows_load_device", "ntt64_fourstep_rows_store_device", "ntt64_fourstep_cols_product_device", "ntt64_twiddle_device", "ntt64_pointwise_device", "ntt64_load_device",
    "ntt64_finalize_device", "ntt64_workspace_bytes", "ntt64_poly_multiply_device",
    "ntt64_workspace_bytes_batched", "ntt64_poly_multiply_batched_device",
    "ntt_transform_u64", "ntt_transform_device_u64", "ntt_transform_device_u32",
    "ntt_io_read_header", "ntt_io_read_in", "ntt_io_check_out", "ntt_io_read_out", "ntt_io_write_out",
    "ntt_profile_enable", "ntt_profile_read",
]


class NttError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"ntt_amd status {status}: {message}")
        self.status = status


class NttLibrary:
    """Typed view of one loaded libnttamd shared object."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). ntt_amd has no CPU fallback.")
        self.path = path
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        L = self.lib
        L.ntt_init.argtypes = [ctypes.c_int]; L.ntt_init.restype = ctypes.c_int
        L.ntt_shutdown.argtypes = []; L.ntt_shutdown.restype = None
        L.ntt_device_count.argtypes = []; L.ntt_device_count.restype = ctypes.c_int
        L.ntt_last_error.argtypes = []; L.ntt_last_error.restype = ctypes.c_char_p
        L.ntt_release_workspace.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.ntt_release_workspace.restype = ctypes.c_int
        L.poly_multiply.argtypes = [c_i32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_int]; L.poly_multiply.restype = None
        L.poly_multiply_u64.argtypes = [c_u64p, c_u64p, c_u64p, ctypes.c_int, ctypes.c_uint64]; L.poly_multiply_u64.restype = None
        L.ntt_poly_multiply_i32.argtypes = [c_i32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_int]; L.ntt_poly_multiply_i32.restype = ctypes.c_int
        L.ntt_poly_multiply_u64.argtypes = [c_u64p, c_u64p, c_u64p, ctypes.c_int, ctypes.c_uint64]; L.ntt_poly_multiply_u64.restype = ctypes.c_int
        L.ntt_poly_multiply_batched.argtypes = [c_u32p, c_u32p, c_u32p, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int]
        L.ntt_poly_multiply_batched.restype = ctypes.c_int
        L.ntt_poly_multiply_batched_u64.argtypes = [c_u64p, c_u64p, c_u64p, ctypes.c_int, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int]
        L.ntt_poly_multiply_batched_u64.restype = ctypes.c_int
        L.ntt_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_size_t]; L.ntt_workspace_bytes.restype = ctypes.c_size_t
        dev_args = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ntt_poly_multiply_batched_device.argtypes = dev_args + [ctypes.c_uint32] + [ctypes.c_size_t] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.ntt_poly_multiply_batched_device.restype = ctypes.c_int
        L.ntt_poly_multiply_batched_device_noalloc.argtypes = dev_args + [ctypes.c_uint32] + [ctypes.c_size_t] * 4 + [ctypes.c_void_p]
        L.ntt_poly_multiply_batched_device_noalloc.restype = ctypes.c_int
        L.ntt_poly_multiply_batched_device_u64.argtypes = dev_args + [ctypes.c_uint64] + [ctypes.c_size_t] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.ntt_poly_multiply_batched_device_u64.restype = ctypes.c_int
        L.ntt_crt_merge_device.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_void_p]
        L.ntt_crt_merge_device.restype = ctypes.c_int
        L.ntt_crt_residue_product_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_size_t] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.ntt_crt_residue_product_device.restype = ctypes.c_int
        L.ntt_crt_prime.argtypes = [ctypes.c_int]; L.ntt_crt_prime.restype = ctypes.c_uint32
        L.ntt_describe_plan.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]; L.ntt_describe_plan.restype = ctypes.c_int
        vp, u64, sz, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
        L.ntt64_transform_device.argtypes = [vp, u64, ci, ci, sz, sz, ci, vp]; L.ntt64_transform_device.restype = ci
        L.ntt64_fourstep_rows_device.argtypes = [vp, u64, ci, ci, sz, u64, ci, vp]; L.ntt64_fourstep_rows_device.restype = ci
        L.ntt64_fourstep_cols_device.argtypes = [vp, u64, ci, ci, ci, sz, sz, ci, vp]; L.ntt64_fourstep_cols_device.restype = ci
        L.ntt64_fourstep_rows_load_device.argtypes = [vp, sz, vp, u64, ci, ci, sz, u64, vp]; L.ntt64_fourstep_rows_load_device.restype = ci
        L.ntt64_fourstep_rows_store_device.argtypes = [vp, vp, u64, ci, ci, sz, u64, vp]; L.ntt64_fourstep_rows_store_device.restype = ci
        L.ntt64_fourstep_cols_product_device.argtypes = [vp, vp, u64, ci, ci, ci, sz, sz, vp]; L.ntt64_fourstep_cols_product_device.restype = ci
        L.ntt64_twiddle_device.argtypes = [vp, u64, ci, sz, sz, u64, u64, ci, ci, ci, vp]; L.ntt64_twiddle_device.restype = ci
        L.ntt64_pointwise_device.argtypes = [vp, vp, sz, u64, ci, vp]; L.ntt64_pointwise_device.restype = ci
        L.ntt64_load_device.argtypes = [vp, sz, sz, vp, sz, sz, u64, ci, vp]; L.ntt64_load_device.restype = ci
        L.ntt64_finalize_device.argtypes = [vp, sz, vp, sz, sz, sz, u64, ci, ci, vp]; L.ntt64_finalize_device.restype = ci
        L.ntt64_workspace_bytes.argtypes = [sz]; L.ntt64_workspace_bytes.restype = sz
        L.ntt64_poly_multiply_device.argtypes = [vp, vp, vp, sz, u64, vp, sz, vp]; L.ntt64_poly_multiply_device.restype = ci
        L.ntt_set_route_policy.argtypes = [ci]; L.ntt_set_route_policy.restype = ci
        L.ntt64_workspace_bytes_batched.argtypes = [sz, sz]; L.ntt64_workspace_bytes_batched.restype = sz
        L.ntt64_poly_multiply_batched_device.argtypes = [vp, vp, vp, sz, u64, sz, sz, sz, sz, vp, sz, vp]
        L.ntt64_poly_multiply_batched_device.restype = ci
        L.ntt_transform_u64.argtypes = [c_u64p, ci, ci, u64]; L.ntt_transform_u64.restype = ci
        L.ntt_transform_device_u64.argtypes = [vp, sz, sz, ci, u64, vp]; L.ntt_transform_device_u64.restype = ci
        L.ntt_transform_device_u32.argtypes = [vp, sz, sz, ci, ctypes.c_uint32, vp]; L.ntt_transform_device_u32.restype = ci
        i64 = ctypes.c_int64
        L.ntt_io_read_header.argtypes = [ctypes.c_char_p, ctypes.POINTER(i64), ctypes.POINTER(u64)]; L.ntt_io_read_header.restype = ci
        L.ntt_io_read_in.argtypes = [ctypes.c_char_p, c_u64p, c_u64p, i64, ci]; L.ntt_io_read_in.restype = ci
        L.ntt_io_check_out.argtypes = [ctypes.c_char_p, c_u64p, i64]; L.ntt_io_check_out.restype = ci
        L.ntt_io_read_out.argtypes = [ctypes.c_char_p, c_u64p, i64]; L.ntt_io_read_out.restype = ci
        L.ntt_io_write_out.argtypes = [ctypes.c_char_p, c_u64p, i64]; L.ntt_io_write_out.restype = ci

    # -- helpers ---------------------------------------------------------
    def check(self, status):
        if status != NTT_OK:
            raise NttError(status, self.lib.ntt_last_error().decode("utf-8", "replace"))

    def last_error(self):
        return self.lib.ntt_last_error().decode("utf-8", "replace")

    def init(self, n_gpus=0):
        self.check(self.lib.ntt_init(int(n_gpus)))

    def shutdown(self):
        self.lib.ntt_shutdown()

    def set_route_policy(self, policy):
        """'auto' (fastest exact route) or 'reference' (four-prime CRT wherever the reference uses it); returns the old one."""
        names = ["auto", "reference"]
        old = self.lib.ntt_set_route_policy(names.index(policy))
        if old < 0:
            raise NttError(-1, self.last_error())
        return names[old]

    def describe_plan(self, n, p):
        buf = ctypes.create_string_buffer(256)
        self.check(self.lib.ntt_describe_plan(int(n), int(p), buf, 256))
        return buf.value.decode()

    # -- host-memory products ---------------------------------------------
    def multiply_i32(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.int32); b = np.ascontiguousarray(b, dtype=np.int32)
        n = a.shape[-1]
        out = np.zeros(2 * n - 1, dtype=np.int32)
        self.check(self.lib.ntt_poly_multiply_i32(a.ctypes.data_as(c_i32p), b.ctypes.data_as(c_i32p), out.ctypes.data_as(c_i32p), n, int(p)))
        return out

    def multiply_u64(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        n = a.shape[-1]
        out = np.zeros(2 * n - 1, dtype=np.uint64)
        self.check(self.lib.ntt_poly_multiply_u64(a.ctypes.data_as(c_u64p), b.ctypes.data_as(c_u64p), out.ctypes.data_as(c_u64p), n, int(p)))
        return out

    def transform_u64(self, a, invert, p):
        """The reference's NTT(a, n, invert, p, 3): natural order in and out (returns a new array)."""
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.check(self.lib.ntt_transform_u64(a.ctypes.data_as(c_u64p), a.size, int(bool(invert)), int(p)))
        return a

    # -- data files of the reference harness (host-only) --------------------
    def read_in_file(self, path, sidecar=False):
        """(n, p, a, b) from `<id>.in`; sidecar=True reuses / refreshes the binary sidecar `<path>.bin`."""
        n, p = ctypes.c_int64(0), ctypes.c_uint64(0)
        path_b = os.fsencode(path)
        self.check(self.lib.ntt_io_read_header(path_b, ctypes.byref(n), ctypes.byref(p)))
        a = np.empty(n.value, dtype=np.uint64); b = np.empty(n.value, dtype=np.uint64)
        self.check(self.lib.ntt_io_read_in(path_b, a.ctypes.data_as(c_u64p), b.ctypes.data_as(c_u64p), n.value, 3 if sidecar else 0))
        return n.value, p.value, a, b

    def read_out_file(self, path, count):
        ab = np.empty(count, dtype=np.uint64)
        self.check(self.lib.ntt_io_read_out(os.fsencode(path), ab.ctypes.data_as(c_u64p), count))
        return ab

    def check_out_file(self, path, ab):
        """True / False as the reference's fCheck decides; raises when the file cannot be opened."""
        ab = np.ascontiguousarray(ab, dtype=np.uint64)
        st = self.lib.ntt_io_check_out(os.fsencode(path), ab.ctypes.data_as(c_u64p), ab.size)
        if st < 0:
            self.check(st)
        return bool(st)

    def write_out_file(self, path, ab):
        ab = np.ascontiguousarray(ab, dtype=np.uint64)
        self.check(self.lib.ntt_io_write_out(os.fsencode(path), ab.ctypes.data_as(c_u64p), ab.size))

    def multiply_batched(self, a, b, p, n_gpus=0):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        batch, n = a.shape
        out = np.empty((batch, 2 * n - 1), dtype=np.uint32)  # every element is written by the call
        self.check(self.lib.ntt_poly_multiply_batched(a.ctypes.data_as(c_u32p), b.ctypes.data_as(c_u32p), out.ctypes.data_as(c_u32p), n, int(p), batch, int(n_gpus)))
        return out

    def multiply_batched_u64(self, a, b, p, n_gpus=0):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        batch, n = a.shape
        out = np.empty((batch, 2 * n - 1), dtype=np.uint64)  # every element is written by the call
        self.check(self.lib.ntt_poly_multiply_batched_u64(a.ctypes.data_as(c_u64p), b.ctypes.data_as(c_u64p), out.ctypes.data_as(c_u64p), n, int(p), batch, int(n_gpus)))
        return out


_product = None


def load_product_library():
    """The HIP library shipped in this package (raises when it has not been built)."""
    global _product
    if _product is None:
        _product = NttLibrary(PRODUCT_LIB)
    return _product
