"""Multi-GPU: environments shard across the GPUs of one node, one process (one engine) per GPU - no PyTorch.

The step path has NO data-path collective: envs never interact (adcraft/gymnasium_kw_env.py:77-103), keywords of one
env interact only through that env's budget and reward, so an env lives on one GPU and every rank steps its own
contiguous block of envs.  Random streams are keyed by the GLOBAL env id (adc_config.env_id_base), so results are
identical at 1/2/4/8 GPUs.  The only exchange is the episode-metric reduction: one RCCL all-reduce(sum) of 3K + 8
doubles over xGMI, performed by the engine itself on its own stream over its device-resident accumulators
(adc_engine_metrics_allreduce; the communicator is brought up by adcraft_amd/comm.py).
"""
import os

import numpy as np

from . import comm as _comm
from . import experiment_metrics as em


def shard_envs(total_envs, world_size, rank):
    """contiguous block partition: returns (num_local_envs, global id of local env 0)"""
    total_envs, world_size, rank = int(total_envs), int(world_size), int(rank)
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total_envs, world_size)
    n = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return n, start


def pack_metric_vector(keyword_profit_cents, keyword_ideal, keyword_ideal_pos, scalars):
    """[sum profit_k (K, cents) | sum ideal_k (K) | sum ideal'_k (K, <= 0 -> 1 per entry) | scalars (8)] as one float64
    vector - the layout adc_engine_metrics_allreduce reduces.  Cents stay exact in float64 up to 2^53."""
    return np.concatenate([np.asarray(keyword_profit_cents, dtype=np.float64), np.asarray(keyword_ideal, dtype=np.float64),
                           np.asarray(keyword_ideal_pos, dtype=np.float64), np.asarray(scalars, dtype=np.float64)])


def unpack_metric_vector(vec, num_keywords):
    K = int(num_keywords)
    return vec[:K], vec[K:2 * K], vec[2 * K:3 * K], vec[3 * K:]


class MetricReducer:
    """The job-wide reduction of one engine's episode metrics.

    backend "rccl" (default): the engine's own RCCL communicator (one GPU per rank); "file": a host-side stand-in for
    ranks that share a GPU or have none (rehearsals and the CPU test of the launcher; ADCRAFT_DIST_BACKEND=file)."""

    def __init__(self, engine=None, rank=None, world_size=None, backend=None):
        r, _, w = _comm.env_rank_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world_size is None else int(world_size)
        self.backend = backend or os.environ.get("ADCRAFT_DIST_BACKEND", "rccl")
        self.engine = engine
        self._file = None
        if self.world > 1:
            if self.backend == "rccl":
                _comm.init_engine_comm(engine, self.rank, self.world)
            elif self.backend == "file":
                self._file = _comm.FileReducer(self.rank, self.world)
            else:
                raise ValueError(f"unknown ADCRAFT_DIST_BACKEND {self.backend!r}")

    def metric_sums(self, ideal_k=None, ideal_pos_k=None):
        """(profit_cents[K], ideal[K], ideal_pos[K], scalars[8]) summed over steps, local envs and ranks"""
        out = self.engine.metrics_allreduce(ideal_k, ideal_pos_k)       # local sums, or the RCCL all-reduce
        if self._file is not None:
            K = len(out[0])
            out = unpack_metric_vector(self._file.allreduce(pack_metric_vector(*out)), K)
        return out

    def allreduce(self, values, op="sum"):
        """sum / max of a small host vector over the ranks (a barrier, the slowest rank's time)"""
        if self._file is not None:
            return self._file.allreduce(values, op)
        if self.engine is not None:
            return self.engine.comm_allreduce(values, op)
        return np.asarray(values, dtype=np.float64)

    def barrier(self):
        self.allreduce([0.0])

    def close(self):
        if self._file is not None:
            self._file.close()


def episode_metrics(profit_cents_k, ideal_k, ideal_pos_k, scalars):
    """AKNCP / NCP (adcraft/experiment_utils/experiment_metrics.py:64-83) from job-wide per-keyword sums.

    The sums pool the envs of the job by KEYWORD INDEX: that is the reference's per-env AKNCP only when every env holds the
    same keyword set (the sharded-replica case); for independent keyword sets use the per-env sums
    (StepEngine.metrics_read_nk).  ideal_pos_k is the ideal with non-positive (step, env, keyword) entries counted as 1
    before summing, as compute_AKNCP does per entry (:71-75)."""
    akncp, ncp = em.akncp_ncp_from_sums(np.asarray(profit_cents_k, dtype=np.float64) / 100.0, ideal_k, ideal_pos_k)
    sc = np.asarray(scalars)
    return dict(AKNCP=akncp, NCP=ncp, profit=float(sc[0]) / 100.0, env_steps=int(sc[1]), episodes=int(sc[2]), truncations=int(sc[3]))


def make_sharded_engine(total_envs, num_keywords, rank, world_size, device_id, seed, **engine_kwargs):
    """this rank's StepEngine for its block of envs (global ids keyed into the random streams)"""
    from .engine import StepEngine
    n, base = shard_envs(total_envs, world_size, rank)
    return StepEngine(n, num_keywords, device_id=device_id, env_id_base=base, seed=seed, **engine_kwargs), n, base
