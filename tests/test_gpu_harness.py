"""The C++ harness (ntt_amd/harness/main.cpp) behaves like the reference main(): same files in, same lines out."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "ntt_amd", "harness", "ntt_harness")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def write_case(dirpath, idx, n, p, a, b, out):
    with open(os.path.join(dirpath, f"{idx}.in"), "w") as f:
        f.write(f"{n} {p}\n" + " ".join(str(int(v)) for v in a) + "\n" + " ".join(str(int(v)) for v in b) + "\n")
    with open(os.path.join(dirpath, f"{idx}.out"), "w") as f:
        f.write(" ".join(str(int(v)) for v in out) + "\n")


def test_harness_reproduces_reference_session(tmp_path):
    assert os.path.exists(HARNESS), "build the harness with __graft_entry__.build()"
    data = tmp_path / "nttdata"; outd = tmp_path / "files"
    data.mkdir(); outd.mkdir()
    z1 = np.load(os.path.join(GOLDEN, "nttdata1.npz")); z4 = np.load(os.path.join(GOLDEN, "ref_crt_n8192.npz"))
    for name in ("0.in", "0.out"):
        (data / name).write_bytes(open(os.path.join(GOLDEN, "nttdata" + name), "rb").read())
    write_case(str(data), 1, int(z1["n"]), int(z1["p"]), z1["a"], z1["b"], z1["out"])
    (data / "2.in").write_bytes(open(os.path.join(GOLDEN, "filetest0.in"), "rb").read())
    (data / "2.out").write_bytes(open(os.path.join(GOLDEN, "filetest0.out"), "rb").read())
    write_case(str(data), 4, int(z4["n"]), int(z4["p"]), z4["a"], z4["b"], z4["out"])   # id 3 is absent on purpose
    r = subprocess.run([HARNESS, str(data), str(outd)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines.count("多项式乘法结果正确") == 4 and "多项式乘法结果错误" not in r.stdout
    assert any(l.startswith("average latency for n = 4 p = 7340033 : ") and l.endswith(" (us) ") for l in r.stdout.splitlines())
    assert any(l.startswith("average latency for n = 131072 p = 7340033 : ") for l in lines)
    assert any(l.startswith("average latency for n = 8192 p = 1337006139375617 : ") for l in lines)
    assert (outd / "0.out").read_text() == "4\n21\n30\n48\n39\n30\n8\n"
    got = np.array((outd / "1.out").read_text().split(), dtype=np.uint64)
    assert np.array_equal(got, z1["out"].astype(np.uint64))
    assert not (outd / "3.out").exists()


def test_harness_reports_a_wrong_result(tmp_path):
    data = tmp_path / "nttdata"; outd = tmp_path / "files"
    data.mkdir(); outd.mkdir()
    (data / "0.in").write_text("4 7340033\n4 1 5 2\n1 5 5 4\n")
    (data / "0.out").write_text("4 21 30 48 39 30 9\n")
    r = subprocess.run([HARNESS, str(data), str(outd)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "多项式乘法结果错误" in r.stdout


def test_harness_with_binary_sidecar_and_phase_timing(tmp_path):
    data = tmp_path / "nttdata"; outd = tmp_path / "files"
    data.mkdir(); outd.mkdir()
    z1 = np.load(os.path.join(GOLDEN, "nttdata2.npz"))
    write_case(str(data), 1, int(z1["n"]), int(z1["p"]), z1["a"], z1["b"], z1["out"])
    env = dict(os.environ, NTT_HARNESS_SIDECAR="1", NTT_HARNESS_VERBOSE="1")
    log = []
    for run in range(2):  # the first run parses the text and leaves 1.in.bin, the second reads it
        r = subprocess.run([HARNESS, str(data), str(outd)], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0 and "多项式乘法结果正确" in r.stdout, r.stdout + r.stderr
        assert "id 1: read " in r.stderr
        assert (data / "1.in.bin").exists()
        log.append(r.stderr)
    got = np.array((outd / "1.out").read_text().split(), dtype=np.uint64)
    assert np.array_equal(got, z1["out"].astype(np.uint64))
    scratch = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(scratch):
        with open(os.path.join(scratch, "harness_timing.txt"), "w") as f:
            f.write("run 1 (text parse):\n" + log[0] + "run 2 (sidecar):\n" + log[1])
