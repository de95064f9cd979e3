"""bench.py prints one JSON line with the agreed keys (a small batch and few steps: the contract, not the speed)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1", "--ramp-ms", "0",
           "--batch", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "poly-mults/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["verified_vs_oracle"] is True and "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "frac_moved", "achieved_moved",
                "dominant_kernel", "kernels", "valu"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the headline fraction prices the whole path: products/s x 96n bytes against the peak; 64n is what really moves
    n = d["config"]["n"]
    assert abs(roof["achieved"] - d["value"] * 96 * n / 1e9) / roof["achieved"] < 1e-9
    assert abs(roof["achieved_moved"] - d["value"] * 64 * n / 1e9) / roof["achieved_moved"] < 1e-9
    assert "static" in roof["traffic_source"] and set(roof["kernels"]) == {"k_fwd_strided", "k_core_mul", "k_inv_strided"}
    assert roof["valu"] is not None and 0 < roof["valu"]["frac"] < 1.5 and roof["valu"]["kernel"] == roof["dominant_kernel"]["kernel"]
    ver = d["verification"]
    assert ver["all_products_evaluation_identity"] is True and ver["rows_checked"] == 8 and ver["oracle_ok"] is True
    # the other single-GPU configurations are timed beside the headline, each with a spot check of its result
    # (the fifth is the n = 2^28 product again in its explicit four-step form, the one that shards over ranks)
    others = d["other_configs"]
    assert len(others) == 5 and all("error" not in o for o in others), others
    assert all(o["verified"] is True and o["poly_mults_per_s"] > 0 for o in others)
    assert others[1]["kernels"].startswith("k64_") and others[2]["kernels"].endswith("k_inv_strided_crt")
    for o in others:     # every configuration prices itself against the HBM roofline, on the model's bytes and on the moved ones
        r = o["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1.0 and 0 < r["frac_moved"] < 1.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "issue" in others[0]["roofline"] and 0 < others[0]["roofline"]["issue"]["frac"] < 1.2
    assert abs(others[4]["ms_per_call"] / others[3]["ms_per_call"] - 1) < 0.10     # four-step form at world 1 = the direct product


def test_more_gpus_than_the_node_has_is_an_error_not_a_smaller_run():
    import torch
    want = torch.cuda.device_count() + 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1", "--ramp-ms", "0",
           "--batch", "2", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert f"--gpus {want} needs {want} GPUs" in r.stderr
