import sys
sys.path.insert(0, ".")
from adcraft_amd.engine import StepEngine
from tests import helpers as H
planes = H.implicit_params(1, 100, seed=3, mean_volume=64)
e = StepEngine(1, 100, seed=5, max_days=1 << 20, loss_threshold=1e12)
e.set_all_params(planes); e.reset(); e.sample_actions(0.3, 1.0, 1e9)
for _ in range(200): e.step_device()
e.synchronize(); e.close()
