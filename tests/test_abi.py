"""The shipped shared object exists, loads without a GPU, and exports every symbol include/ntt_amd.h declares."""
import os
import re

from ntt_amd._binding import EXPORTS, PRODUCT_LIB, NttLibrary

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ntt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b((?:ntt_|ntt64_|poly_multiply)\w*)\s*\(", text))


def test_header_and_binding_agree():
    assert declared_symbols() == set(EXPORTS)


def test_library_loads_and_exports_every_declared_symbol():
    lib = NttLibrary(PRODUCT_LIB)
    for name in declared_symbols():
        assert getattr(lib.lib, name) is not None, name


def test_no_gpu_means_an_error_not_a_fallback():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        return
    lib = NttLibrary(PRODUCT_LIB)
    a = np.arange(4, dtype=np.uint64)
    out = np.zeros(7, dtype=np.uint64)
    import ctypes
    u64p = ctypes.POINTER(ctypes.c_uint64)
    st = lib.lib.ntt_poly_multiply_u64(a.ctypes.data_as(u64p), a.ctypes.data_as(u64p), out.ctypes.data_as(u64p), 4, 7340033)
    assert st == -3 and "no usable HIP device" in lib.last_error()
    assert not out.any()


def test_product_library_has_no_link_to_the_oracle():
    import subprocess
    deps = subprocess.run(["ldd", PRODUCT_LIB], capture_output=True, text=True).stdout
    assert "oracle" not in deps and "cputwin" not in deps


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/ntt_amd.h compiles as C99 and a C program that binds the two drop-in symbols links against the library."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    src = tmp_path / "bind.c"
    src.write_text(
        '#include "ntt_amd.h"\n'
        "int main(void) {\n"
        "    int a[4] = {4, 1, 5, 2}, b[4] = {1, 5, 5, 4}, ab[8] = {0};\n"
        "    if (ntt_poly_multiply_i32(a, b, ab, 4, 7340033) == NTT_OK) return ab[0] == 4 ? 0 : 1;\n"
        "    return ntt_last_error()[0] ? 3 : 2;   /* no GPU: an error with a message, never a made-up result */\n"
        "}\n")
    exe = tmp_path / "bind"
    lib_dir = os.path.dirname(PRODUCT_LIB)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", str(src), "-I", os.path.join(ROOT, "include"), "-L", lib_dir,
                        "-lnttamd", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    import torch
    assert run.returncode == (0 if torch.cuda.is_available() else 3), (run.returncode, run.stderr)
