"""GPU box: what env groups do to steps that are JOINED every time (a policy kernel, a fetch or a synchronize between two steps)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
N, K, mean_volume, cvr, no_vol_prob, _ = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)

def timed(fn, steps=100, warm=30):
    for _ in range(warm):
        fn()
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    e.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for groups in (0, 1):
    row = [f"groups {groups}:"]
    e = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e12, drift_enabled=True, auto_reset=True)
    e.set_env_groups(groups)
    e.set_all_params(planes)
    e.reset()
    e.bid_curves_build(2048)
    e.metrics_enable(True)
    e.agent_init(1.0, None)
    def loop_agent():
        e.agent_step(100000.0); e.ideal_step(fetch=False); e.step_device()
    row.append(f"agent+ideal+step {timed(loop_agent):.4f}")
    def loop_agent_only():
        e.agent_step(100000.0); e.step_device()
    row.append(f"agent+step {timed(loop_agent_only):.4f}")
    e.close()
    for budget in (1e9, 1000.0):
        e = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e12, auto_reset=True)
        e.set_env_groups(groups)
        e.set_all_params(planes)
        e.reset()
        e.sample_actions(0.3, 1.0, budget)
        e.metrics_enable(True)
        def loop_sync():
            e.step_device(); e.synchronize()
        row.append(f"step+sync@{budget:g} {timed(loop_sync):.4f}")
        def loop_resample():
            e.sample_actions(0.3, 1.0, budget); e.step_device()
        row.append(f"sample+step@{budget:g} {timed(loop_resample):.4f}")
        row.append(f"run_days(fixed)@{budget:g} {timed(lambda: e.run_days('fixed', 1, budget=budget, graph=False)):.4f}")
        bids, bud = np.full((N, K), 0.8, np.float32), np.full(N, budget, np.float32)
        row.append(f"host step@{budget:g} {timed(lambda: e.step(bids, bud), 40, 10):.4f}")
        row.append(f"back-to-back@{budget:g} {timed(e.step_device):.4f}")
        e.close()
    print("  ".join(row), flush=True)
