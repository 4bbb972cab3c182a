#!/usr/bin/env python3
"""GPU box: reset() of a default-constructor (EXPLICIT) vector env - sample_random_keywords for every env - by the three samplers:
the reference's seeded recipe in a host loop, numpy for all envs at once, and on the device (k_generate_explicit_keywords)."""
import sys
import time

sys.path.insert(0, ".")
from adcraft_amd.vector_env import BiddingSimulationVectorEnv  # noqa: E402

for N, K in ((4096, 64), (16384, 256)):
    for sampler in ("reference", "vectorised", "device"):
        if sampler == "reference" and N > 4096:
            continue
        vec = BiddingSimulationVectorEnv(N, num_keywords=K, param_sampler=sampler)
        vec.reset(seed=1)
        t0 = time.perf_counter()
        vec.reset(seed=2)
        vec.engine.synchronize()
        print(f"default-constructor env, {N} x {K}, param_sampler={sampler:10s}: reset(seed) {1e3 * (time.perf_counter() - t0):9.2f} ms", flush=True)
        vec.close()
