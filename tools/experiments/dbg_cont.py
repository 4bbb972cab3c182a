import sys
sys.path.insert(0, ".")
import numpy as np
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
N, K = 8, 256
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=128, cvr=0.8)
e = StepEngine(N, K, seed=1729)
e.set_all_params(planes); e.reset(); e.bid_curves_build(2048)
n, lst, iv = e.bid_curves_contenders()
print("contender counts: min", n.min(), "median", np.median(n), "mean", n[n < 65535].mean() if (n < 65535).any() else None, "max", n.max(), "overflow share", (n == 65535).mean())
ir, cpc = e.bid_curves_fetch()
k = 3
print("kw", k, "count", n[0, k], lst[0, k, :min(n[0, k], 48)])
u = np.unique(np.stack([ir[0, k], cpc[0, k]]), axis=1).shape[1]
print("distinct lines", u)
