"""bench.py's multi-rank control flow without GPUs: `--gpus N` started as one plain process fans out to N ranks (the
place of the reference's `mpiexec -np N`, qsub_mpi.sh:16), the launcher form used for N > 1 agrees with it, and a
mismatch between `--gpus` and the started ranks is an error, never a mislabelled line.  The ranks run the test-only
CPU twin over gloo (`--rehearse-cpu`), and the line says so."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN = os.path.join(ROOT, "tests", "cpu_twin", "libnttamd_cputwin.so")
SMALL = ["--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--size", "2048", "--batch", "4", "--no-cpu-baseline",
         "--rehearse-cpu", "--lib", TWIN]


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-1000:]
    return json.loads(lines[0])


def clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}


def test_gpus_flag_alone_starts_that_many_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True,
                       timeout=300, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "batch-sharded x2" and "rehearsal" in d
    assert d["verified_vs_oracle"] is True and d["verification"]["all_ranks_ok"] is True
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6   # whole-job rate over both ranks
    # the ranks attest that they talk to one another, and the same processes time the other configurations across the ranks:
    # batch-sharded small and large-modulus products, the CRT work split by (prime, row) with its all-to-all, the four-step product
    assert d["rccl"] == {"backend": "gloo", "world_size": 2, "all_reduce_of_ones": 2.0, "devices": 0}
    others = d["other_configs"]
    assert len(others) == 4 and all(o["n_gpus"] == 2 and o["verified"] is True and "roofline" in o for o in others), others
    assert [o["scaling"] for o in others] == ["weak", "weak", "strong", "strong"]
    assert others[2]["exchange"]["collective"] == "all_to_all_single" and others[3]["exchange"]["per_product"] == 3
    assert all(len(o["rank_ms"]) == 2 and o["rank_ms"][0] <= o["rank_ms"][1] for o in others)


def test_launcher_form_and_rank_mismatch():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")]
    r = subprocess.run(base + ["--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=300, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json_line(r.stdout)["n_gpus"] == 2
    # two ranks started, one GPU asked for: refuse
    base[base.index("--master-port") + 1] = str(free_port())
    r = subprocess.run(base + ["--gpus", "1"] + SMALL, capture_output=True, text=True, timeout=300, env=clean_env(), cwd=ROOT)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 1 but the launcher started 2" in r.stderr


def test_headline_line_survives_other_configurations_that_do_not_finish():
    # a deadline shorter than any workload: every rank gives the other configurations up, rank 0 still prints the headline
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")]
    r = subprocess.run(base + ["--gpus", "2", "--others-deadline-s", "0.001"] + SMALL, capture_output=True, text=True,
                       timeout=300, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["verified_vs_oracle"] is True and d["value"] > 0
    assert len(d["other_configs"]) == 1 and "not finished within" in d["other_configs"][0]["error"]
    assert "other configurations abandoned" in r.stderr


def test_eight_ranks_time_every_configuration():
    # the rank count of a whole node: the four other configurations across eight ranks (one prime per pair of ranks in the
    # CRT split, a 64 x 128 four-step matrix), every verdict reduced over the ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + SMALL, capture_output=True, text=True,
                       timeout=600, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["verified_vs_oracle"] is True and d["verification"]["all_ranks_ok"] is True
    assert d["rccl"]["world_size"] == 8 and d["rccl"]["all_reduce_of_ones"] == 8.0
    others = d["other_configs"]
    assert len(others) == 4 and all(o["n_gpus"] == 8 and o["verified"] is True and len(o["rank_ms"]) == 2 for o in others), others
