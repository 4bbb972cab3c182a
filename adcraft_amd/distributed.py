"""Multi-GPU: environments shard across the GPUs of one node, one process per GPU.

The step path has NO data-path collective: envs never interact (adcraft/gymnasium_kw_env.py:77-103), keywords of
one env interact only through that env's budget and reward, so an env lives on one GPU and every rank steps its
own contiguous block of envs.  Random streams are keyed by the GLOBAL env id (adc_config.env_id_base), so results
are identical at 1/2/4/8 GPUs.  The only exchange is the episode-metric reduction: one all-reduce(sum) of a
small vector - RCCL over xGMI when the process group is "nccl" (ROCm), gloo in the CPU tests.
"""
import numpy as np

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


def pack_metric_vector(keyword_profit_cents, keyword_ideal, scalars):
    """[sum_env profit_k (K, cents) | sum_env ideal_k (K, dollars) | scalars (8)] as one float64 vector.
    Cents stay exact in float64 up to 2^53."""
    return np.concatenate([np.asarray(keyword_profit_cents, dtype=np.float64),
                           np.asarray(keyword_ideal, dtype=np.float64), np.asarray(scalars, dtype=np.float64)])


def unpack_metric_vector(vec, num_keywords):
    K = int(num_keywords)
    return vec[:K], vec[K:2 * K], vec[2 * K:]


def all_reduce_sum(vec, group=None, device=None):
    """the single collective of the path.  With an initialised torch.distributed process group the vector is
    summed over ranks (on `device` if given - "cuda" + nccl backend = RCCL over xGMI); without one it is returned
    unchanged (single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(vec, dtype=np.float64)
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def episode_metrics(local_keyword_profit_cents, local_keyword_ideal, local_scalars, group=None, device=None):
    """global AKNCP / NCP (adcraft/experiment_utils/experiment_metrics.py:64-83) from per-rank per-keyword sums
    over (steps x local envs).  Returns dict(AKNCP, NCP, profit, env_steps, episodes, truncations)."""
    K = len(local_keyword_profit_cents)
    total = all_reduce_sum(pack_metric_vector(local_keyword_profit_cents, local_keyword_ideal, local_scalars), group, device)
    profit_c, ideal, sc = unpack_metric_vector(total, K)
    akncp, ncp = em.akncp_ncp_from_sums(profit_c / 100.0, ideal)
    return dict(AKNCP=akncp, NCP=ncp, profit=float(sc[0]) / 100.0, env_steps=int(sc[1]), episodes=int(sc[2]),
                truncations=int(sc[3]))


def make_sharded_engine(total_envs, num_keywords, rank, world_size, device_id, seed, **engine_kwargs):
    """this rank's StepEngine for its block of envs (global ids keyed into the random streams)"""
    from .engine import StepEngine
    n, base = shard_envs(total_envs, world_size, rank)
    return StepEngine(n, num_keywords, device_id=device_id, env_id_base=base, seed=seed, **engine_kwargs), n, base
