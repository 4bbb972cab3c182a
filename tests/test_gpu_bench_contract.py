"""-m gpu: bench.py as the driver runs it - one JSON line with the contract's keys, sane values."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), *args], text=True, cwd=ROOT, timeout=600)
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--gpus", "1", "--steps", "6", "--warmup", "2", "--cpu-seconds", "1.0")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value is whole-job keyword-steps/s: units / time
    assert d["value"] == pytest.approx(4096 * 256 * 6 / (d["ms_per_step"] * 6e-3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert r["algorithmic_bytes_per_launch"] == 4096 * 256 * 56 + 26 * 4096 and r["launches"] == 6
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.01           # the kernel runs inside the timed region
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 100 * c["value"]
    assert 0 < d["episode_metric"]["NCP"] < 2


def test_bench_other_configs_run():
    d = _run("--config", "cfg5", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    assert "drift True" in d["config"]["workload"] and d["roofline"]["algorithmic_bytes_per_launch"] == 2048 * 1024 * 68 + 26 * 2048
    d = _run("--budget", "50", "--steps", "4", "--warmup", "2", "--no-cpu-baseline")
    assert d["roofline"]["kernel"].startswith("k_step_exact_rows")
