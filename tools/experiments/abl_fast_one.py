import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
N, K = 4096, 256
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
print(f"{sys.argv[1]:>18}: fast kernel {kms[0]/40*1e3:.1f} us", flush=True)
eng.close()
