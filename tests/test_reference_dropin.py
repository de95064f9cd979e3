"""The reference's own program on the library.  oracle/Makefile compiles /root/reference/main_naive.cc unmodified, weakens
its two multiply functions in the object file and links the strong forwarding definitions of oracle/dropin_glue.cc (the
binding a maintainer would write) plus libnttamd.so over them: main(), fRead, fCheck and fWrite are the reference's, the
product is the library's (call sites main_naive.cc:253-257).  The reference reads "/nttdata/<id>.in" (main_naive.cc:15); the
glue redirects that prefix to $NTT_DATA_DIR.  Ids 0-3 must print the reference's "correct" line; id 4's data is absent from
the reference tree, so the program re-runs id 3's operands against a missing file and prints "wrong", as the reference
itself does.  Without a GPU the binary is run against the CPU twin to prove the link and the call path; with one, against
the shipped library."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "main_naive_dropin")
GOLDEN = os.path.join(ROOT, "tests", "golden")
OK, WRONG = "多项式乘法结果正确", "多项式乘法结果错误"


def make_data_dir(tmp_path):
    d = tmp_path / "nttdata"; d.mkdir(); (tmp_path / "files").mkdir()
    shutil.copy(os.path.join(GOLDEN, "nttdata0.in"), d / "0.in"); shutil.copy(os.path.join(GOLDEN, "nttdata0.out"), d / "0.out")
    outs = {}
    for i in (1, 2, 3):
        z = np.load(os.path.join(GOLDEN, f"nttdata{i}.npz"))
        (d / f"{i}.in").write_text(f"{int(z['n'])} {int(z['p'])}\n" + " ".join(map(str, z["a"].tolist())) + "\n" +
                                   " ".join(map(str, z["b"].tolist())) + "\n")
        (d / f"{i}.out").write_text(" ".join(map(str, z["out"].tolist())) + "\n")
        outs[i] = z["out"]
    return d, outs


def run_and_check(tmp_path, env_extra):
    d, outs = make_data_dir(tmp_path)
    env = dict(os.environ, NTT_DATA_DIR=str(d), **env_extra)
    r = subprocess.run([BINARY], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    verdicts = [l for l in r.stdout.splitlines() if l in (OK, WRONG)]
    assert verdicts == [OK, OK, OK, OK, WRONG], r.stdout          # ids 0-3 right; id 4 has no data (see module docstring)
    assert "average latency for n = 131072 p = 469762049" in r.stdout
    for i in (1, 2, 3):                                            # fWrite's files equal the reference's expected outputs
        got = np.loadtxt(tmp_path / "files" / f"{i}.out", dtype=np.uint64)
        assert np.array_equal(got, outs[i].astype(np.uint64))


needs_binary = pytest.mark.skipif(not os.path.exists(BINARY), reason="oracle/_ref/main_naive_dropin not built (needs /root/reference)")


@needs_binary
def test_reference_object_calls_the_library_symbol():
    nm = subprocess.run(["nm", "-D", BINARY], capture_output=True, text=True).stdout
    assert " U poly_multiply_u64" in nm                            # the product comes from the shared library ...
    nm_all = subprocess.run(["nm", "-C", BINARY], capture_output=True, text=True).stdout
    assert "fRead(" in nm_all and "fCheck(" in nm_all and "fWrite(" in nm_all   # ... the harness from the reference's object
    ldd = subprocess.run(["ldd", BINARY], capture_output=True, text=True).stdout
    assert "libnttamd.so" in ldd


@needs_binary
def test_reference_main_runs_on_the_cpu_twin(tmp_path, twin):
    lib_dir = tmp_path / "lib"; lib_dir.mkdir()
    os.symlink(twin.path, lib_dir / "libnttamd.so")
    run_and_check(tmp_path, {"LD_LIBRARY_PATH": str(lib_dir)})


@needs_binary
@pytest.mark.gpu
def test_reference_main_runs_on_the_gpu_library(tmp_path, gpu_lib):
    run_and_check(tmp_path, {})
