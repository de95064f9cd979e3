import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle_binding import Reference
    return Reference()


_product_libraries = []   # every product library a fixture has loaded (the twin and the GPU library are separate images)


@pytest.fixture(params=["auto", "reference"])
def route(request):
    """Runs the test once per route policy: the fastest route for large NTT-friendly moduli (direct 64-bit field) and
    the reference's own (four-prime CRT).  Request it after the library fixture."""
    for lib in _product_libraries:
        lib.set_route_policy(request.param)
    yield request.param
    for lib in _product_libraries:
        lib.set_route_policy("auto")


@pytest.fixture(scope="session")
def twin():
    """CPU twin of the GPU library (same sources over a HIP stand-in); test-only."""
    from ntt_amd._binding import NttLibrary
    path = os.path.join(ROOT, "tests", "cpu_twin", "libnttamd_cputwin.so")
    if not os.path.exists(path):
        pytest.skip("CPU twin not built (make -C tests/cpu_twin)")
    lib = NttLibrary(path)
    _product_libraries.append(lib)
    return lib


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from ntt_amd._binding import load_product_library
    lib = load_product_library()
    lib.init(0)
    _product_libraries.append(lib)
    return lib
