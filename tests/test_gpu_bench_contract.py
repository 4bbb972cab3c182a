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
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "also"):
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
    _kernel_time_fits_the_step(d)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["single_thread"]["cores"] == 1 and 0 < c["single_thread"]["value"] <= c["value"] * 1.5
    assert c["reference_python"]["value"] > 0 and "provenance" in c["reference_python"]
    assert d["value"] > 100 * c["value"]
    assert 0 < d["episode_metric"]["NCP"] < 2
    if r["traffic"] is not None:
        assert r["traffic_source"]                      # a number not measured in this process says where it is from
    # the largest single-GPU config and the per-GPU shards of the 8-GPU configs, measured in the same run
    a = d["also"]
    assert set(a) == {"cfg3", "cfg4", "cfg5"}
    assert a["cfg3"]["steps"] >= 50 and a["cfg3"]["roofline"]["algorithmic_bytes_per_launch"] == 16384 * 1024 * 56 + 26 * 16384
    assert a["cfg3"]["value"] == pytest.approx(16384 * 1024 / (a["cfg3"]["ms_per_step"] * 1e-3), rel=1e-6)
    assert a["cfg3"]["roofline"]["frac"] > d["roofline"]["frac"]          # the sparse config is the HBM-heavy one


def _kernel_time_fits_the_step(d):
    """the kernel time (untimed pass of the same length, the batch as one launch) cannot exceed the time of a step scheduled the same
    way by more than noise; a step run as env groups on several streams may be shorter than that, and the line says so"""
    r, g = d["roofline"], d["env_groups"]
    assert g["profiled_pass"] == 1 and r["envs_per_launch"]["timed_region"] * g["timed_region"] == r["envs_per_launch"]["profiled_pass"]
    assert r["launches_in_timed_region"] == (d["steps"] if g["timed_region"] == 1 else 1 + (d["steps"] - 1) * g["timed_region"])
    if g["timed_region"] == 1:
        assert r["kernel_ms"] <= d["ms_per_step"] * 1.05
    else:
        assert r["kernel_ms"] <= g["ms_per_step_as_one_group"] * 1.05
        assert d["ms_per_step"] <= g["ms_per_step_as_one_group"] * 1.05 and "BELOW" in g["note"]


def test_driver_shaped_run_times_a_region_without_event_records():
    """what the driver runs (--steps 20 --warmup 5): exactly 20 launches in the timed region, no event record among them; the
    kernel time comes from the separate, untimed, equally long pass and cannot exceed the step time by more than noise"""
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-also")
    r = d["roofline"]
    assert d["steps"] == 20 and d["warmup"] == 5
    assert r["launches_in_timed_region"] == 1 + 19 * d["env_groups"]["timed_region"] and r["event_records_in_timed_region"] == 0 and r["launches"] == 20
    assert "untimed pass" in r["kernel_ms_method"]
    _kernel_time_fits_the_step(d)


def test_bench_other_configs_run():
    d = _run("--config", "cfg5", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    assert "drift True" in d["config"]["workload"] and d["roofline"]["algorithmic_bytes_per_launch"] == 2048 * 1024 * 68 + 26 * 2048
    assert "also" not in d
    d = _run("--budget", "50", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-also")
    assert "k_step_exact_rows" in d["roofline"]["kernel"]


def test_bench_two_ranks_on_one_gpu_with_the_stand_in_reduction():
    """`--gpus 2` starts two rank processes; on a one-GPU box both engines share the device, so the reduction goes
    through the file stand-in (RCCL refuses two ranks on one device; the RCCL call itself is covered at world size 1 by
    test_gpu_device_resident.py).  Everything else - sharding by global env id, the per-episode reduction inside the timed
    region, the slowest rank's time, rank 0's line - is the N > 1 path as the driver runs it."""
    env = dict(os.environ, ADCRAFT_DIST_BACKEND="file")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "61", "--warmup", "2",
                                   "--config", "cfg5"], text=True, cwd=ROOT, timeout=900, env=env)
    d = json.loads([ln for ln in out.strip().splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 2048 * 1024 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["episode_metric"]["env_steps"] == 2 * 2048 * 61
