"""The exact double-precision field of the 64-bit kernels (ntt_amd/csrc/ntt_field64.h) compiled for the CPU and driven
with millions of random operands at and inside its range limits, every result compared with 128-bit integer arithmetic
(tests/fieldd_stress.cpp).  Same IEEE-754 operations as on the GPU: multiply, add, fused multiply-add, no contraction."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_primitive_is_exact_on_ten_million_random_operands(tmp_path):
    exe = tmp_path / "fieldd_stress"
    flags = ["-O2", "-std=c++17", "-ffp-contract=off"]
    with open("/proc/cpuinfo") as f:
        hw_fma = " fma " in f.read().replace("\n", " ")
    if hw_fma:
        flags.append("-mfma")          # hardware fused multiply-add; otherwise libm's (exact as well, slower)
    r = subprocess.run(["g++", *flags, "-I", os.path.join(ROOT, "ntt_amd", "csrc"), "-o", str(exe),
                        os.path.join(ROOT, "tests", "fieldd_stress.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    samples = 2000000 if hw_fma else 200000
    r = subprocess.run([str(exe), str(samples)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"FIELDD STRESS OK {5 * samples}" in r.stdout, r.stdout[-500:]
