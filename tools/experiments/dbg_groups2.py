"""GPU box: does the env groups' overlap survive engines created one after another, and the null stream coming into being in between?"""
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
seq = sys.argv[1].split(",") if len(sys.argv) > 1 else ["plain"] * 4 + ["walk"] + ["plain"] * 4
for i, variant in enumerate(seq):
    eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, 1000.0)
    if variant == "walk":
        eng.walk_stats(reset=True)
    ms = device_ms(eng)
    print(i, variant, f"{ms:.4f} ms/step groups={eng.env_groups()}", flush=True)
    eng.close()
