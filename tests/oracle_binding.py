"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when present, to the reference
programs compiled as shared objects (oracle/_ref).  Test infrastructure only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
u64p = ctypes.POINTER(ctypes.c_uint64)
i32p = ctypes.POINTER(ctypes.c_int32)


def _u64(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle`")
        self.lib = L = ctypes.CDLL(path)
        L.oracle_pow.argtypes = [ctypes.c_uint64] * 3; L.oracle_pow.restype = ctypes.c_uint64
        for name in ("oracle_ntt_multiply", "oracle_crt_ntt_multiply", "oracle_poly_multiply_u64", "oracle_schoolbook_u64"):
            f = getattr(L, name); f.argtypes = [u64p, u64p, u64p, ctypes.c_int, ctypes.c_uint64]; f.restype = None
        L.oracle_ntt_multiply_openmp.argtypes = [u64p, u64p, u64p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        L.oracle_ntt_multiply_openmp.restype = None
        L.oracle_openmp_max_threads.restype = ctypes.c_int
        L.oracle_ntt.argtypes = [u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]; L.oracle_ntt.restype = None
        L.oracle_ntt_base4.argtypes = [u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]; L.oracle_ntt_base4.restype = None
        L.oracle_reverse.argtypes = [u64p, ctypes.c_int, ctypes.c_int]; L.oracle_reverse.restype = None
        L.oracle_poly_multiply_schoolbook.argtypes = [i32p, i32p, i32p, ctypes.c_int, ctypes.c_int]
        L.oracle_poly_multiply_schoolbook.restype = None
        L.oracle_barrett64_reduce.argtypes = [ctypes.c_uint64] * 2; L.oracle_barrett64_reduce.restype = ctypes.c_uint64
        L.oracle_barrett62_reduce.argtypes = [ctypes.c_uint64] * 2; L.oracle_barrett62_reduce.restype = ctypes.c_uint64
        L.oracle_mont64fields_multiply.argtypes = [ctypes.c_uint64] * 3; L.oracle_mont64fields_multiply.restype = ctypes.c_uint64
        L.oracle_read_in.argtypes = [ctypes.c_char_p, u64p, u64p, ctypes.POINTER(ctypes.c_int), u64p, ctypes.c_int]
        L.oracle_read_in.restype = ctypes.c_int
        L.oracle_check_out.argtypes = [ctypes.c_char_p, u64p, ctypes.c_int]; L.oracle_check_out.restype = ctypes.c_int
        L.oracle_write_out.argtypes = [ctypes.c_char_p, u64p, ctypes.c_int]; L.oracle_write_out.restype = ctypes.c_int
        L.oracle_fill_random.argtypes = [u64p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]; L.oracle_fill_random.restype = None

    def _mul(self, fn, a, b, p):
        a, b = _u64(a), _u64(b)
        n = a.size
        # the transform-based functions share the reference's limit: len = 2n must be a power of two
        # (main_naive.cc:139-146); other sizes are checked against the schoolbook product instead
        if fn is not self.lib.oracle_schoolbook_u64 and n & (n - 1):
            raise ValueError("the transform oracle needs n to be a power of two; use schoolbook_u64")
        out = np.zeros(2 * n - 1, dtype=np.uint64)
        fn(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p), n, int(p))
        return out

    def ntt_multiply(self, a, b, p): return self._mul(self.lib.oracle_ntt_multiply, a, b, p)
    def crt_ntt_multiply(self, a, b, p): return self._mul(self.lib.oracle_crt_ntt_multiply, a, b, p)
    def poly_multiply_u64(self, a, b, p): return self._mul(self.lib.oracle_poly_multiply_u64, a, b, p)
    def schoolbook_u64(self, a, b, p): return self._mul(self.lib.oracle_schoolbook_u64, a, b, p)

    def ntt_multiply_openmp(self, a, b, p, threads=0):
        a, b = _u64(a), _u64(b)
        n = a.size
        out = np.zeros(2 * n - 1, dtype=np.uint64)
        self.lib.oracle_ntt_multiply_openmp(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p), n, int(p), int(threads))
        return out

    def ntt_base4(self, a, invert, p, g=3):
        a = _u64(a).copy()
        assert a.size >= 1 and (a.size - 1) & a.size == 0 and (a.size.bit_length() - 1) % 2 == 0, "n must be a power of four"
        self.lib.oracle_ntt_base4(a.ctypes.data_as(u64p), a.size, int(bool(invert)), int(p), int(g))
        return a

    def ntt(self, a, invert, p, g=3):
        a = _u64(a).copy()
        self.lib.oracle_ntt(a.ctypes.data_as(u64p), a.size, int(bool(invert)), int(p), int(g))
        return a

    def schoolbook_i32(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.int32); b = np.ascontiguousarray(b, dtype=np.int32)
        n = a.size
        out = np.zeros(2 * n - 1, dtype=np.int32)
        self.lib.oracle_poly_multiply_schoolbook(a.ctypes.data_as(i32p), b.ctypes.data_as(i32p), out.ctypes.data_as(i32p), n, int(p))
        return out

    def fill_random(self, count, p, seed):
        out = np.zeros(count, dtype=np.uint64)
        self.lib.oracle_fill_random(out.ctypes.data_as(u64p), count, int(p), int(seed))
        return out

    def read_in(self, path, cap=300000):
        a = np.zeros(cap, dtype=np.uint64); b = np.zeros(cap, dtype=np.uint64)
        n = ctypes.c_int(0); p = ctypes.c_uint64(0)
        rc = self.lib.oracle_read_in(path.encode(), a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), ctypes.byref(n), ctypes.byref(p), cap)
        if rc != 0:
            raise IOError(path)
        return n.value, p.value, a[:n.value].copy(), b[:n.value].copy()

    def check_out(self, path, ab, n):
        ab = _u64(ab)
        return self.lib.oracle_check_out(path.encode(), ab.ctypes.data_as(u64p), n)

    def write_out(self, path, ab, n):
        ab = _u64(ab)
        return self.lib.oracle_write_out(path.encode(), ab.ctypes.data_as(u64p), n)


class Reference:
    """The reference's own functions, compiled unmodified from /root/reference into oracle/_ref
    (C++-mangled symbols).  Available wherever oracle/_ref/*.so exist."""

    def __init__(self):
        self.naive = self._load("libref_naive.so")
        self.openmp = self._load("libref_openmp.so")
        self.barrett = self._load("libref_barrett.so")
        sig = [u64p, u64p, u64p, ctypes.c_int, ctypes.c_uint64]
        if self.naive is not None:
            for s in ("_Z12NTT_multiplyPmS_S_im", "_Z23CRT_NTT_multiply_serialPmS_S_im"):
                f = getattr(self.naive, s); f.argtypes = sig; f.restype = None
        if self.barrett is not None:
            f = getattr(self.barrett, "_Z12NTT_multiplyPmS_S_im"); f.argtypes = sig; f.restype = None
        if self.openmp is not None:
            f = getattr(self.openmp, "_Z21NTT_multiply_parallelPmS_S_imi"); f.argtypes = sig + [ctypes.c_int]; f.restype = None
            f = getattr(self.openmp, "_Z23CRT_NTT_multiply_openmpPmS_S_im"); f.argtypes = sig; f.restype = None
            g = getattr(self.openmp, "_Z22init_global_crt_valuesv"); g.argtypes = []; g.restype = None
            g()

    @staticmethod
    def _load(name):
        path = os.path.join(REF_DIR, name)
        return ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL) if os.path.exists(path) else None

    @property
    def available(self):
        return self.naive is not None

    @staticmethod
    def _call(fn, a, b, p, *extra):
        a, b = _u64(a), _u64(b)
        n = a.size
        out = np.zeros(2 * n, dtype=np.uint64)
        fn(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p), n, int(p), *extra)
        return out[:2 * n - 1]

    def ntt_multiply(self, a, b, p): return self._call(self.naive._Z12NTT_multiplyPmS_S_im, a, b, p)
    def crt_ntt_multiply(self, a, b, p): return self._call(self.naive._Z23CRT_NTT_multiply_serialPmS_S_im, a, b, p)
    def barrett_ntt_multiply(self, a, b, p): return self._call(self.barrett._Z12NTT_multiplyPmS_S_im, a, b, p)
    def openmp_ntt_multiply(self, a, b, p, threads=8):
        return self._call(self.openmp._Z21NTT_multiply_parallelPmS_S_imi, a, b, p, int(threads))
    def openmp_crt_multiply(self, a, b, p): return self._call(self.openmp._Z23CRT_NTT_multiply_openmpPmS_S_im, a, b, p)
