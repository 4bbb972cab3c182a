"""GPU box: env groups under the stream scheme ADCRAFT_EXP_STREAM_MODE, engines created one after another, the null stream coming into
being before the third (walk_stats); metric mode and auto-reset as in bench.py"""
import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
def device_ms(eng, steps=100):
    for _ in range(3):
        for _ in range(6):
            eng.step_device()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step_device()
    eng.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
out = []
for i, variant in enumerate(["plain", "plain", "walk", "plain", "plain"]):
    for cfg, budget in (("cfg2", 1e9), ("cfg2", 1000.0), ("cfg5", 1e9)):
        N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
        planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
        eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e15, auto_reset=True, drift_enabled=drift)
        eng.set_all_params(planes)
        eng.reset()
        eng.sample_actions(0.30, 1.00, budget)
        eng.metrics_enable(True)
        if variant == "walk":
            eng.walk_stats(reset=True)
        out.append(f"{variant[0]}:{cfg}@{budget:g}={device_ms(eng):.4f}(g{eng.env_groups()})")
        eng.close()
print("  ".join(out), flush=True)
