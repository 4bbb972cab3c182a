#!/usr/bin/env python3
"""Context number (build container only, needs /root/reference): keyword-steps/s of the REFERENCE's own Python loop
(simulate_epoch_of_bidding_on_campaign, unmodified, same in-memory stand-ins as tools/gen_golden.py; the Rust volume
sampler replaced by its law on numpy) on one core, for the dense cfg2 keyword law at K = 100.
Usage: python tools/time_reference_python.py [mean_volume] [K] [days]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

mean_volume = float(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
days = int(sys.argv[3]) if len(sys.argv) > 3 else 5
G.install_standins()
from adcraft import bidding_simulation as b, gymnasium_kw_utils as u  # noqa: E402
from adcraft.experiment_utils import experiment_quantiles as eq  # noqa: E402

rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(1)))
cfg, _ = G.quant_cfg(eq, mean_volume, 0.8, None)
_, params = u.sample_implicit_keywords_from_quantile_dfs(K, rng, cfg)
kws = []
for p in params:
    vol = (float(p[0][0]), float(p[0][1]))
    kw, _ = u.generate_implicit_keyword_from_params(vol, p[1], 1.0 / p[2], p[3], p[4], p[5], p[6], rng)
    kw.volume_sampler = (lambda m, s: (lambda: int(np.floor(max(rng.normal(m, max(s, 1e-12)), 0.0) + 0.5))))(vol[0], vol[1])
    kws.append(kw)
bids = [float(x) for x in np.around(rng.uniform(0.3, 1.0, K), 2)]
b.simulate_epoch_of_bidding_on_campaign(kws, bids, 1e9)
t0 = time.perf_counter()
for _ in range(days):
    b.simulate_epoch_of_bidding_on_campaign(kws, bids, 1e9)
dt = (time.perf_counter() - t0) / days
print(f"reference Python loop, 1 core, K={K}, mean_volume={mean_volume:g}: {dt * 1e3:.1f} ms per env-day = {K / dt:.0f} keyword-steps/s "
      f"({K * mean_volume / dt:.3g} auctions/s)")
