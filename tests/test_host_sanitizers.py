"""The library's host code and kernel indexing under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only): the
sources over the CPU stand-in plus a driver that sweeps the entry points against the oracle, built by
`make -C tests/sanitize` (about two minutes, therefore not built by the suite itself)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "tests", "sanitize", "host_sanitize")


def test_entry_point_sweep_is_clean_under_asan_and_ubsan():
    if not os.path.exists(BINARY):
        pytest.skip("sanitizer build absent (make -C tests/sanitize)")
    sources = [os.path.join(ROOT, "ntt_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "ntt_amd", "csrc"))]
    if any(os.path.getmtime(s) > os.path.getmtime(BINARY) for s in sources):
        pytest.skip("sanitizer build older than the sources (make -C tests/sanitize)")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([BINARY], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "HOST SANITIZE OK" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr
