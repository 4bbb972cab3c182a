"""Multi-process CPU tests (world size 2) of the N > 1 host path: env sharding, the hand-over of the communicator id from
rank 0 to the other ranks (adcraft_amd/comm.py - no PyTorch in the product), the metric vector and its reduction, AKNCP /
NCP from reduced sums, and bench.py starting its own ranks.

The engine itself needs a GPU (the RCCL all-reduce behind the C ABI is exercised at world size 1 by
tests/test_gpu_device_resident.py).  Here the per-rank "local sums" come from the CPU oracle run on each rank's shard with
GLOBAL env keys - which also checks that sharded results equal the unsharded ones - and the reduction runs through the
file-based stand-in, cross-checked against a gloo all-reduce of the same vector (torch.distributed is used by this TEST
only, as an independent second implementation)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
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
    ideal = (planes[6].astype(np.float64) - 0.9).sum(axis=0) * steps          # a per-keyword "ideal", negative for some keywords
    ideal_pos = np.where(planes[6].astype(np.float64) - 0.9 <= 0, 1.0, planes[6].astype(np.float64) - 0.9).sum(axis=0) * steps
    return prof, ideal, ideal_pos, sc


def _worker(rank, world, port, job, K, steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ADCRAFT_JOB_ID=job)
    import torch
    import torch.distributed as dist
    from adcraft_amd import comm, distributed as D
    # 1. the communicator id travels from rank 0 to everybody (here: 128 recognisable bytes instead of an RCCL id)
    uid = comm.exchange_bytes(rank, world, lambda: bytes((7 * i + 3) % 256 for i in range(128)))
    # 2. this rank's shard and its local sums
    n, base = D.shard_envs(16, world, rank)
    prof, ideal, ideal_pos, sc = _local_sums(n, base, K, steps)
    # 3. the reduction (file stand-in) against gloo on the same vector
    red = comm.FileReducer(rank, world)
    vec = D.pack_metric_vector(prof, ideal, ideal_pos, sc)
    total = red.allreduce(vec)
    slowest = red.allreduce([float(rank + 1)], op="max")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.from_numpy(vec.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    same_as_gloo = bool(np.array_equal(t.numpy(), total))
    m = D.episode_metrics(*D.unpack_metric_vector(total, K))
    q.put((rank, n, base, m, uid, same_as_gloo, float(slowest[0])))
    dist.barrier()
    dist.destroy_process_group()
    red.close()


def test_shard_bounds():
    from adcraft_amd import distributed as D
    for total, world in [(16, 2), (65536, 8), (10, 3), (7, 8)]:
        parts = [D.shard_envs(total, world, r) for r in range(world)]
        assert sum(n for n, _ in parts) == total
        assert [b for _, b in parts] == list(np.cumsum([0] + [n for n, _ in parts[:-1]]))
    with pytest.raises(ValueError):
        D.shard_envs(4, 2, 2)


def test_world2_id_exchange_and_metric_reduction_equal_single_process():
    K, steps, world = 24, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    job = f"pytest{os.getpid()}"
    procs = [ctx.Process(target=_worker, args=(r, world, port, job, K, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [(r[1], r[2]) for r in res] == [(8, 0), (8, 8)]
    assert res[0][4] == res[1][4] == bytes((7 * i + 3) % 256 for i in range(128))      # the id arrived intact
    assert res[0][5] and res[1][5]                     # the stand-in reduction == gloo's all-reduce, bit for bit
    assert res[0][6] == res[1][6] == 2.0               # max over ranks
    assert res[0][3] == res[1][3]                      # every rank holds the same reduced metric
    assert not [f for f in os.listdir("/tmp") if f"_{job}." in f]          # nothing left behind
    # single-process reference over all 16 envs
    sys.path.insert(0, ROOT)
    from adcraft_amd import distributed as D, experiment_metrics as em
    prof, ideal, ideal_pos, sc = _local_sums(16, 0, K, steps)
    single = D.episode_metrics(prof, ideal, ideal_pos, sc)
    got = res[0][3]
    for k in ("profit", "env_steps", "episodes", "truncations"):      # cents and counts: exact
        assert single[k] == got[k], k
    for k in ("AKNCP", "NCP"):      # the f64 ideal sums are added in a different order across ranks
        assert got[k] == pytest.approx(single[k], rel=1e-12)
    akncp, ncp = em.akncp_ncp_from_sums(prof / 100.0, ideal, ideal_pos)
    assert single["AKNCP"] == akncp and single["NCP"] == ncp and single["env_steps"] == 16 * steps
    # the per-entry replacement matters: applying "<= 0 -> 1" to the sums instead gives another AKNCP here
    assert em.akncp_ncp_from_sums(prof / 100.0, ideal)[0] != akncp


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` outside a launcher starts two fresh rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment), before anything touches a GPU, and prints rank 0's line.  --rehearse replaces the GPU
    work by a token workload so that the launch, the id hand-over and the reduction run on a machine without a GPU."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["rehearsal"]["ranks_seen"] == [0, 1] and line["rehearsal"]["id_bytes"] == 128
    # two communicators are brought up back to back (as cfg4 then cfg5 are): each hand-over delivers its own id
    assert line["rehearsal"]["bring_ups"] == 2 and line["rehearsal"]["ids_distinct"] and line["rehearsal"]["second_reduction"] == [2.0, 2.0]
    assert "cfg4" in line["config"]["workload"]
    # a failing rank fails the launch
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse", "--fail-rank", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
    assert "rank 1 stderr" in bad.stderr and "fails on request" in bad.stderr          # what the failing rank said is relayed
    # a job that outlives --launch-timeout is ended, not waited for
    slow = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse",
                           "--launch-timeout", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert slow.returncode != 0 and "launch-timeout" in slow.stderr
    assert not [f for f in os.listdir("/tmp") if f.startswith("adcraft_comm_") and "_bench" in f]          # nothing left behind
