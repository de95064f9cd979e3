"""The library's data-file scanner/formatter (ntt_io_*) against an independent pure-Python reader, the
reference's own files, and the byte format fWrite produces (main_naive.cc:13-67).  Host-only: no GPU."""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, read_in_file, read_out_file
from ntt_amd._binding import NttError, NttLibrary, PRODUCT_LIB


@pytest.fixture(scope="module")
def lib():
    return NttLibrary(PRODUCT_LIB)


@pytest.mark.parametrize("name", ["nttdata0", "filetest0", "filetest1", "filetest2", "filetest3"])
def test_reference_files_parse_like_the_plain_reader(lib, name):
    path = os.path.join(GOLDEN, name + ".in")
    n, p, a, b = lib.read_in_file(path)
    n2, p2, a2, b2 = read_in_file(path)
    assert (n, p) == (n2, p2) and np.array_equal(a, a2) and np.array_equal(b, b2)
    want = read_out_file(path[:-3] + ".out", n)
    assert np.array_equal(lib.read_out_file(path[:-3] + ".out", 2 * n - 1), want)
    assert lib.check_out_file(path[:-3] + ".out", want) is True
    bad = want.copy(); bad[-1] ^= 1
    assert lib.check_out_file(path[:-3] + ".out", bad) is False


def test_write_is_byte_identical_to_the_reference_format(lib, tmp_path):
    vals = np.array([0, 1, 9, 10, 998244352, 2**32, 1337006139375616, 2**64 - 1], dtype=np.uint64)
    out = tmp_path / "x.out"
    lib.write_out_file(str(out), vals)
    assert out.read_bytes() == b"".join(b"%d\n" % int(v) for v in vals)
    assert np.array_equal(lib.read_out_file(str(out), vals.size), vals)
    # the reference's own output file re-emitted from its parsed values is unchanged up to line endings
    ref = os.path.join(GOLDEN, "filetest0.out")
    n = 4096
    lib.write_out_file(str(out), read_out_file(ref, n))
    assert out.read_bytes().split() == open(ref, "rb").read().split()


def test_large_file_round_trip_and_tolerant_white_space(lib, tmp_path):
    rng = np.random.default_rng(3)
    n, p = 131072, 1337006139375617
    a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
    path = tmp_path / "big.in"
    with open(path, "wb") as f:
        f.write(b"%d %d\r\n" % (n, p))
        f.write(b" ".join(b"%d" % int(v) for v in a) + b"\n\n\t")
        f.write(b"\n".join(b"%d" % int(v) for v in b))  # no trailing newline
    n2, p2, a2, b2 = lib.read_in_file(str(path))
    assert (n2, p2) == (n, p) and np.array_equal(a2, a) and np.array_equal(b2, b)


def test_sidecar_is_used_only_while_the_text_file_is_unchanged(lib, tmp_path):
    path = tmp_path / "s.in"
    path.write_bytes(b"2 7340033\n1 2\n3 4\n")
    n, p, a, b = lib.read_in_file(str(path), sidecar=True)
    bin_path = tmp_path / "s.in.bin"
    assert bin_path.exists() and (n, p, a.tolist(), b.tolist()) == (2, 7340033, [1, 2], [3, 4])
    # corrupt the payload of the sidecar: a sidecar-enabled read now returns the corrupt values (proof it is used) ...
    raw = bytearray(bin_path.read_bytes()); raw[-8:] = (99).to_bytes(8, "little"); bin_path.write_bytes(bytes(raw))
    assert lib.read_in_file(str(path), sidecar=True)[3].tolist() == [3, 99]
    # ... a plain read ignores it, and after the text file changes the sidecar is stale and rebuilt from the text
    assert lib.read_in_file(str(path))[3].tolist() == [3, 4]
    path.write_bytes(b"2 7340033\n5 6\n7 8\n")
    os.utime(path, ns=(1, 1))
    n, p, a, b = lib.read_in_file(str(path), sidecar=True)
    assert (a.tolist(), b.tolist()) == ([5, 6], [7, 8])
    assert lib.read_in_file(str(path), sidecar=True)[2].tolist() == [5, 6]


@pytest.mark.parametrize("text", [b"", b"4\n", b"2 7 1 2 3\n", b"2 7 1 x 3 4\n", b"2 7 1 2 3 99999999999999999999999\n", b"0 7\n", b"2 7 1 2 3 4.5\n"])
def test_malformed_inputs_are_errors(lib, tmp_path, text):
    path = tmp_path / "bad.in"
    path.write_bytes(text)
    with pytest.raises(NttError) as e:
        lib.read_in_file(str(path))
    assert e.value.status == -5


def test_missing_files(lib, tmp_path):
    with pytest.raises(NttError):
        lib.read_in_file(str(tmp_path / "none.in"))
    with pytest.raises(NttError):
        lib.check_out_file(str(tmp_path / "none.out"), np.zeros(3, dtype=np.uint64))
    short = tmp_path / "short.out"; short.write_bytes(b"1\n2\n")
    assert lib.check_out_file(str(short), np.array([1, 2, 3], dtype=np.uint64)) is False
