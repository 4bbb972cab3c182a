"""GPU box: the closed loop by env groups (0 = the engine's choice): Python loop of agent_step / ideal_step / step_device, and run_days"""
import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
N, K, mean_volume, cvr, no_vol_prob, _ = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
for groups in (1, 0, 1, 0):
    e = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e12, drift_enabled=True, auto_reset=True)
    e.set_env_groups(groups)
    e.set_all_params(planes)
    e.reset()
    e.bid_curves_build(2048)
    e.metrics_enable(True)
    e.agent_init(1.0, None)
    out = []
    def timed(fn, steps=100, warm=30):
        for _ in range(warm):
            fn()
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        e.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    def a():
        e.agent_step(100000.0); e.ideal_step(fetch=False); e.step_device()
    out.append(f"agent+ideal+step {timed(a):.4f} (g{e.env_groups()})")
    def b():
        e.agent_step(100000.0); e.step_device()
    out.append(f"agent+step {timed(b):.4f}")
    def c():
        e.ideal_step(fetch=False); e.policy_oracle(100000.0); e.step_device()
    out.append(f"ideal+oracle+step {timed(c):.4f}")
    def d():
        e.sample_actions(0.3, 1.0, 1000.0); e.step_device()
    out.append(f"sample+step@1000 {timed(d):.4f}")
    for pol in ("zero_margin", "oracle", "fixed"):
        e.run_days(pol, 30, budget=100000.0, graph=False)
        e.synchronize()
        t0 = time.perf_counter()
        e.run_days(pol, 100, budget=100000.0, graph=False)
        e.synchronize()
        out.append(f"run_days({pol}) {(time.perf_counter() - t0) / 100 * 1e3:.4f}")
    print(f"groups {groups}: " + "  ".join(out), flush=True)
    e.close()
