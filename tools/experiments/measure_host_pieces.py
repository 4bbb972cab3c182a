#!/usr/bin/env python3
"""where a host-in / host-out step of 4096 x 256 goes: staging copy, transfers, kernels (single engine and 4 shards)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adcraft_amd import synthetic  # noqa: E402
from adcraft_amd.engine import ShardedStepEngine, StepEngine  # noqa: E402

N, K = 4096, 256
planes = synthetic.implicit_keyword_planes(N, K, seed=1)
bids = np.full((N, K), 0.8, np.float32)
budget = np.full(N, 1e6, np.float32)


def t(f, n=30):
    for _ in range(3):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


e = StepEngine(N, K, seed=1)
e.set_all_params(planes)
e.reset()
print(f"numpy staging copy of the bids (4 MB):        {t(lambda: e._bids_stage.__setitem__(Ellipsis, bids)):.3f} ms")
print(f"single engine, step() incl. staging:          {t(lambda: e.step(bids, budget, copy=False)):.3f} ms")


def dev_only():
    e.step_device()
    e.synchronize()


print(f"single engine, device step + sync:            {t(dev_only):.3f} ms")
print(f"single engine, fetch() only (21 MB D2H):      {t(lambda: e.fetch(copy=False)):.3f} ms")
e.close()
for shards in (2, 4, 6):
    s = ShardedStepEngine(N, K, shards=shards, seed=1)
    s.set_all_params(planes)
    s.reset()
    print(f"{shards} engines, step() incl. staging:               {t(lambda: s.step(bids, budget, copy=False)):.3f} ms")

    def no_stage():
        for (p, b0, b1), o in zip(s._each(), s._outs):
            p.step_async(s._bids_stage[b0:b1].ctypes.data, s._budget_stage[b0:b1].ctypes.data, o)
        for p in s.parts:
            p.wait()
    print(f"{shards} engines, actions already in the pinned buffer: {t(no_stage):.3f} ms")
    s.close()

# how much of a sharded step is host-side enqueueing (API calls) and how much is waiting for the device
s = ShardedStepEngine(N, K, shards=4, seed=1)
s.set_all_params(planes)
s.reset()
enq = wait = 0.0
for i in range(33):
    t0 = time.perf_counter()
    for (p, b0, b1), o, (pb, pg) in zip(s._each(), s._outs, s._in_ptrs):
        p.step_async(pb, pg, o)
    t1 = time.perf_counter()
    for p in s.parts:
        p.wait()
    t2 = time.perf_counter()
    if i >= 3:
        enq, wait = enq + (t1 - t0), wait + (t2 - t1)
print(f"4 engines: enqueueing {enq / 30 * 1e3:.3f} ms, then waiting {wait / 30 * 1e3:.3f} ms")
s.close()
