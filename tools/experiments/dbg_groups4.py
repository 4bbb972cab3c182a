import sys, time, traceback
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
for cfg in sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5"]:
    N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg]
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
    eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e15, auto_reset=True, drift_enabled=drift)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, 1e9)
    eng.metrics_enable(True)
    eng.ideal_profit(2048)
    try:
        t0 = time.perf_counter()
        eng.step_device()
        eng.synchronize()
        print(cfg, "first step ok", f"{(time.perf_counter() - t0) * 1e3:.2f} ms", "groups", eng.env_groups(), flush=True)
        for _ in range(20):
            eng.step_device()
        eng.synchronize()
        print(cfg, "ok groups", eng.env_groups(), flush=True)
    except Exception:
        traceback.print_exc()
    eng.close()
