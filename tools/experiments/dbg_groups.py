import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
def device_ms(eng, steps=60):
    for _ in range(3):
        for _ in range(4):
            eng.step_device()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step_device()
    eng.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for variant in ("plain", "walk_stats", "metrics", "autoreset"):
    for budget in (1e9, 1000.0):
        kw = dict(max_days=1 << 30, loss_threshold=1e15)
        if variant == "autoreset":
            kw = dict(max_days=60, loss_threshold=1e15, auto_reset=True)
        eng = StepEngine(N, K, seed=1729, **kw)
        eng.set_all_params(planes)
        eng.reset()
        eng.sample_actions(0.30, 1.00, budget)
        if variant == "walk_stats":
            eng.walk_stats(reset=True)
        if variant == "metrics":
            eng.metrics_enable(True)
        ms = device_ms(eng)
        print(variant, budget, f"{ms:.4f} ms/step groups={eng.env_groups()} kernel={eng.step_kernel_name()}", flush=True)
        eng.close()
