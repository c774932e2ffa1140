"""bench.py's output contract: ONE JSON line with the driver's keys plus `roofline` and `cpu_baseline`."""
import json
import os
import py_compile
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_bench_and_entry_compile():
    py_compile.compile(BENCH, doraise=True)
    py_compile.compile(os.path.join(ROOT, "__graft_entry__.py"), doraise=True)


def test_bench_refuses_to_run_without_a_gpu(gpu_present):
    if gpu_present:
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, BENCH, "--steps", "30", "--warmup", "5", "--preheat-ms", "50", "--cpu-seconds", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "Mpoints/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["achieved"] > 1000                                   # sanity: > 1 TB/s on an MI355X
    # value and the per-launch figure describe the same steps
    assert abs(d["value"] - 8 * 1280 * 720 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mpoints/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
