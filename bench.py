#!/usr/bin/env python3
"""bench.py - throughput of the BiddingSimulation step engine on N MI355X GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU.  A "step" is one pass of the hot path (one
BiddingSimulation.step for every resident env) over device-resident synthetic actions.
Workload at every N: BASELINE.json configs[1] per GPU (4096 envs x 256 keywords, dense stationary
keyword law) - envs shard with no data-path collective, so scaling is weak; for N>1 the episode
metric vector is all-reduced over RCCL (torch.distributed "nccl") every max_days steps.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_U = {False: 56, True: 68}   # 36 B read + 20 B written per keyword-step (+12 B drift write-back)
BYTES_PER_ENV = 26


def cpu_baseline(cfg_name, planes, K, seconds=12.0):
    """the CPU oracle (a C restatement of the reference's loops, oracle/adcraft_oracle.c) on this host's cores,
    on a bounded sample of the same workload"""
    from oracle import capi as orc
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    n_envs = 8 * cores
    o = orc.OracleEngine(n_envs, K, threads=cores)
    o.params[:] = planes[:, :n_envs]
    o.key[:] = np.arange(1, n_envs + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    bids = o.sample_bids(0.3, 1.0)
    o.step(bids, 1.0e9)                      # warm
    t0 = time.perf_counter()
    steps = 0
    auctions = 0
    while time.perf_counter() - t0 < seconds:
        out = o.step(bids, 1.0e9)
        auctions += int(out["volumes"].sum())
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": n_envs * K * steps / dt, "unit": "keyword-steps/s", "cores": cores, "kind": "port",
            "auctions_per_s": auctions / dt,
            "sample": f"{n_envs} envs x {K} keywords of {cfg_name}, {steps} steps, {dt:.1f} s, OpenMP over envs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", help="cfg2 (metric config), cfg3, cfg4, cfg5")
    ap.add_argument("--budget", type=float, default=1.0e9, help="per-env daily budget in dollars (default: non-binding)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    backend = os.environ.get("ADCRAFT_DIST_BACKEND", "nccl")     # "gloo" lets the N>1 path be rehearsed on one GPU
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_device = "cuda" if backend == "nccl" else None

    from adcraft_amd import _ffi, synthetic
    from adcraft_amd.engine import StepEngine

    N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[args.config]
    max_days = 60
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729 + rank, mean_volume=mean_volume, cvr=cvr,
                                               no_vol_prob=no_vol_prob)
    n_dev = max(_ffi.device_count(), 1)
    eng = StepEngine(N, K, device_id=local_rank % n_dev, seed=1729, env_id_base=rank * N, max_days=max_days,
                     loss_threshold=1.0e12, drift_enabled=drift, auto_reset=True)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, args.budget)  # actions resident in HBM before the timed region
    eng.metrics_enable(True)

    from adcraft_amd import distributed as D, experiment_metrics as em
    # stationary keywords: the ideal (max expected) profit per keyword is constant over the episode
    ideal_k = eng.ideal_profit(2048).sum(axis=0)      # (with drift on this is the episode-start value)

    def metric_allreduce(steps_done):
        """the single collective of the path: [sum profit_k | sum ideal_k | scalars], RCCL over xGMI for N>1"""
        kp, sc = eng.metrics_read()
        vec = D.pack_metric_vector(kp, ideal_k * steps_done, sc)
        if dist is not None:
            vec = D.all_reduce_sum(vec, device=red_device)
        return vec

    def barrier():
        eng.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step_device()
    metric_allreduce(args.warmup)       # also brings up the RCCL communicator outside the timed region
    barrier()
    eng.metrics_reset()
    eng.profile_enable(True)
    eng.profile_read()
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        eng.step_device()
        if dist is not None and (s + 1) % max_days == 0:
            metric_allreduce(s + 1)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eng.profile_read()
    eng.profile_enable(False)
    totals = metric_allreduce(args.steps)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        units = float(world) * N * K * args.steps
        b_alg = BYTES_PER_U[drift] * N * K + BYTES_PER_ENV * N            # algorithmic bytes per launch (one GPU)
        names = ("k_step_implicit_fast", "k_step_exact_rows (+ step tail)", "k_metric_accumulate")
        dom = int(np.argmax(kernel_ms))              # the dominant kernel of the step
        k_ms = kernel_ms[dom] / max(launches, 1)
        achieved = b_alg / (k_ms * 1e-3) / 1e9 if launches else None
        traffic, valu = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pmc_cfg = json.load(open(pmc)).get(args.config, {})
                traffic = pmc_cfg.get("hbm_bytes_per_launch") if dom == 0 else None
                valu = pmc_cfg.get("valu") if dom == 0 else None
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/sec (envs\u00d7keywords auctions/s)",
            "value": units / elapsed,
            "unit": "keyword-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32 cents + f32 (Philox4x32-10 u32)", "data": "synthetic",
            "config": {"workload": f"{args.config}: {N} envs x {K} keywords per GPU, IMPLICIT keywords, "
                                   f"mean_volume {mean_volume}, cvr {cvr}, no_vol_prob {no_vol_prob}, drift {drift}, "
                                   f"budget {'non-binding' if args.budget >= 1e8 else args.budget}, {max_days}-step episodes with auto-reset",
                       "envs_per_gpu": N, "keywords": K, "parallelism": f"env-sharded x{world}"},
            "env_steps_per_s": float(world) * N * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": names[dom], "kernel_ms": k_ms, "launches": int(launches),
                         "all_kernels_ms": {n: m / max(launches, 1) for n, m in zip(names, kernel_ms)},
                         "algorithmic_bytes_per_launch": b_alg},
        }
        if valu and launches:
            # why the HBM fraction is low on dense configs: the kernel is VALU-issue-bound (static PMC counts of this
            # workload from profiles/pmc_traffic.json, divided by the live kernel time)
            line["roofline"]["issue"] = {"valu_wave_instructions_per_launch": valu["wave_instructions_per_launch"],
                                         "valu_wave_instructions_per_s": valu["wave_instructions_per_launch"] / (k_ms * 1e-3),
                                         "valu_lane_instructions_per_auction": valu.get("valu_lane_instructions_per_auction")}
        profit_c, ideal, sc = D.unpack_metric_vector(totals, K)
        akncp, ncp = em.akncp_ncp_from_sums(profit_c / 100.0, ideal)
        line["episode_metric"] = {"AKNCP": akncp, "NCP": ncp, "profit_dollars": float(sc[0]) / 100.0,
                                  "env_steps": int(sc[1]), "episodes": int(sc[2]),
                                  "note": "synthetic uniform bids, not a trained agent"}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.config, planes, K, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
