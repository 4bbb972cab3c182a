"""Multi-process CPU test (gloo, world_size 2) of the N>1 host path: env sharding, the metric vector and the one
all-reduce, AKNCP/NCP from reduced sums.  The engine itself needs a GPU; the per-rank "local sums" here come
from the CPU oracle run on each rank's shard with GLOBAL env keys, which also checks that sharded results equal
the unsharded ones."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keys(n, base):
    ids = np.arange(base, base + n, dtype=np.uint64)
    return (ids + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)


def _local_sums(n, base, K, steps):
    from oracle import capi as orc
    from tests import helpers as H
    planes = H.implicit_params(16, K, seed=77)[:, base:base + n]
    o = orc.OracleEngine(n, K)
    o.params[:] = planes
    o.key[:] = _keys(n, base)
    prof = np.zeros(K, dtype=np.int64)
    sc = np.zeros(8, dtype=np.int64)
    for _ in range(steps):
        out = o.step(o.sample_bids(0.3, 1.0), 1e9)
        prof += (out["revenue_cents"] - out["cost_cents"]).sum(axis=0)
        sc[0] += int(np.rint(out["reward"] * 100).sum())
        sc[1] += n
    ideal = np.abs(planes[6].astype(np.float64)).sum(axis=0) * steps      # any positive per-keyword denominator
    return prof, ideal, sc


def _worker(rank, world, port, K, steps, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adcraft_amd import distributed as D
    n, base = D.shard_envs(16, world, rank)
    prof, ideal, sc = _local_sums(n, base, K, steps)
    m = D.episode_metrics(prof, ideal, sc)
    q.put((rank, n, base, m))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from adcraft_amd import distributed as D
    for total, world in [(16, 2), (65536, 8), (10, 3), (7, 8)]:
        parts = [D.shard_envs(total, world, r) for r in range(world)]
        assert sum(n for n, _ in parts) == total
        assert [b for _, b in parts] == list(np.cumsum([0] + [n for n, _ in parts[:-1]]))
    with pytest.raises(ValueError):
        D.shard_envs(4, 2, 2)


def test_world2_metric_allreduce_equals_single_process():
    K, steps, world = 24, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [(r[1], r[2]) for r in res] == [(8, 0), (8, 8)]
    assert res[0][3] == res[1][3]                      # every rank holds the same reduced metric
    # single-process reference over all 16 envs
    sys.path.insert(0, ROOT)
    from adcraft_amd import distributed as D, experiment_metrics as em
    prof, ideal, sc = _local_sums(16, 0, K, steps)
    single = D.episode_metrics(prof, ideal, sc)         # no process group here: identity reduction
    got = res[0][3]
    for k in ("profit", "env_steps", "episodes", "truncations"):      # cents and counts: exact
        assert single[k] == got[k], k
    for k in ("AKNCP", "NCP"):      # the f64 ideal sums are added in a different order across ranks
        assert got[k] == pytest.approx(single[k], rel=1e-12)
    akncp, ncp = em.akncp_ncp_from_sums(prof / 100.0, ideal)
    assert single["AKNCP"] == akncp and single["NCP"] == ncp and single["env_steps"] == 16 * steps
