import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
K = 256
for N in (1280, 2560, 3840, 4096, 4480, 5120, 6400, 3840, 4096):
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729)
    eng = StepEngine(N, K, seed=1729, max_days=60, loss_threshold=1e15, auto_reset=True)
    eng.set_all_params(planes); eng.reset(); eng.sample_actions(0.30, 1.00, 1e9); eng.metrics_enable(True)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(20): eng.step_device()
        eng.synchronize()
    eng.profile_enable(True); eng.profile_read()
    for _ in range(40): eng.step_device()
    eng.synchronize()
    kms, launches = eng.profile_read()
    eng.profile_enable(False)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(60): eng.step_device()
        eng.synchronize()
        best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
    print(f"N={N} rounds={N/1280:.2f} fast kernel {kms[0]/40*1e3:.1f} us  ({kms[0]/40*1e3/N*1280:.2f} us per 1280 tiles)  step {best*1e3:.1f} us groups={eng.env_groups()}", flush=True)
    eng.close()
