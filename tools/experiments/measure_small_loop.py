"""Launch-bound regime: per-day time of the device-resident loop at small env counts, one call per kernel vs
adc_engine_run_days (pairs of days replayed from a captured hipGraph)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from adcraft_amd.engine import StepEngine
from tests import helpers as H
for N, K in [(1, 100), (16, 100), (256, 100)]:
    planes = H.implicit_params(N, K, seed=3, mean_volume=64)
    e = StepEngine(N, K, seed=5, max_days=60, loss_threshold=1e12, drift_enabled=True)
    e.set_all_params(planes); e.reset()
    e.bid_curves_build(2048); e.metrics_enable(True); e.agent_init(1.0, None)
    def loop(n):
        for _ in range(n):
            e.agent_step(100000.0); e.ideal_step(fetch=False); e.step_device()
        e.synchronize()
    loop(5)
    t0 = time.perf_counter(); loop(60); dt = time.perf_counter() - t0
    e.run_days("zero_margin", 6, 100000.0, graph=True); e.synchronize()
    t0 = time.perf_counter(); e.run_days("zero_margin", 60, 100000.0); e.synchronize(); dtg = time.perf_counter() - t0
    e.sample_actions(0.3, 1.0, 1e9)
    def loop2(n):
        for _ in range(n): e.step_device()
        e.synchronize()
    loop2(5)
    t0 = time.perf_counter(); loop2(200); dt2 = (time.perf_counter() - t0) / 200
    e.run_days("fixed", 6); e.synchronize()
    t0 = time.perf_counter(); e.run_days("fixed", 200); e.synchronize(); dt2g = (time.perf_counter() - t0) / 200
    print(f"N={N} K={K}: closed-loop day {dt/60*1e6:.1f} us, from the hipGraph {dtg/60*1e6:.1f} us; "
          f"bare device step {dt2*1e6:.1f} us, from the hipGraph {dt2g*1e6:.1f} us")
    e.close()
