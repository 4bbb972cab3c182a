#!/usr/bin/env python3
"""bench.py - throughput of the BiddingSimulation step engine on N MI355X GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; rank 0 prints ONE JSON line.
  * launched by `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this process is one
    rank; otherwise, with --gpus N > 1, it starts N fresh rank processes itself - before anything touches a GPU - collects
    rank 0's line and fails if any rank fails.  No PyTorch anywhere: the one collective (the episode-metric all-reduce) is
    the engine's own RCCL call behind the C ABI (adc_engine_metrics_allreduce), brought up by adcraft_amd/comm.py.
  * A "step" is one pass of the hot path (one BiddingSimulation.step for every resident env) over device-resident
    synthetic actions.
  * Workload.  N = 1: BASELINE.json configs[1] (cfg2: 4096 envs x 256 keywords, dense stationary law) is `value`; the
    largest single-GPU config (cfg3: 16384 x 1024, sparse volume) and the per-GPU shards of the 8-GPU configs are measured in
    the same run and reported under "also".  N > 1: every GPU holds the BASELINE configs[3] shard (cfg4: 65536 x 1024 over
    8 GPUs = 8192 x 1024 per GPU), envs shard with no data-path collective ("weak" scaling), the metric vector is
    all-reduced once per 60-step episode inside the timed region; configs[4] (cfg5: drift) is under "also".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9      # 256 CUs x 4 SIMD-32 x 2.4 GHz: one wave64 instruction per 2 cycles per SIMD
# ... which only packed / pure add-fma streams reach.  A wave64 instruction of the classes these kernels are made of (compares,
# converts, min / max, integer multiplies, v_cndmask, mbcnt, anything VOP3 with a literal) occupies its SIMD for 4 cycles
# (profiles/r02_issue_rates.md; the round-4 ablations price the sparse kernel at 4.0-4.3 cycles per instruction): the rate this
# instruction mix can reach is 16 lanes per SIMD and cycle
VALU_PEAK_4CYCLE_LANE_OPS = 256 * 4 * 16 * 2.4e9
BYTES_PER_U = {False: 56, True: 68}   # 36 B read + 20 B written per keyword-step (+12 B drift write-back)
BYTES_PER_ENV = 26
MAX_DAYS = 60
# the three intervals between the engine's HIP events; the IMPLICIT kernels add their metric sums in their own output phase, so
# the third interval holds no kernel here (what it shows is the cost of recording an event pair)
KERNEL_NAMES = ("k_step_implicit_fast", "k_tail_or_flag + the budget-exact kernels (k_step_click_walk, k_step_exact_rows, k_step_rest_of_day, k_rest_walk)",
                "(no kernel: event-record overhead)")
# the reference's own Python loop, unmodified, on this keyword law: measured in the BUILD container (tools/time_reference_python.py,
# one Xeon core @ 2.1 GHz, stand-ins for the two modules that cannot be imported there) - never on the GPU box, where
# the reference does not exist
REFERENCE_PYTHON = {"value": 631.0, "unit": "keyword-steps/s", "cores": 1,
                    "provenance": "tools/time_reference_python.py in the build container (158 ms per 100-keyword env-day); DESIGN.md section 6"}


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(cfg_name, planes, K, seconds):
    """the CPU oracle (a C restatement of the reference's loops, oracle/adcraft_oracle.c) on this host, on a bounded sample
    of the same workload: one thread, then all cores"""
    from oracle import capi as orc
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))

    def timed(threads, n_envs, budget_s):
        o = orc.OracleEngine(n_envs, K, threads=threads)
        o.params[:] = planes[:, :n_envs]
        o.key[:] = np.arange(1, n_envs + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        bids = o.sample_bids(0.3, 1.0)
        o.step(bids, 1.0e9)                      # warm
        t0 = time.perf_counter()
        steps = auctions = 0
        while time.perf_counter() - t0 < budget_s:
            out = o.step(bids, 1.0e9)
            auctions += int(out["volumes"].sum())
            steps += 1
        dt = time.perf_counter() - t0
        return n_envs * K * steps / dt, auctions / dt, steps, dt

    one, one_a, s1, d1 = timed(1, 8, seconds * 0.4)
    allc, all_a, s2, d2 = timed(cores, 8 * cores, seconds * 0.6)
    return {"value": allc, "unit": "keyword-steps/s", "cores": cores, "kind": "port", "auctions_per_s": all_a,
            "single_thread": {"value": one, "cores": 1, "auctions_per_s": one_a,
                              "sample": f"8 envs x {K} keywords of {cfg_name}, {s1} steps, {d1:.1f} s"},
            "sample": f"{8 * cores} envs x {K} keywords of {cfg_name}, {s2} steps, {d2:.1f} s, OpenMP over envs",
            "reference_python": REFERENCE_PYTHON}


# ---------------------------------------------------------------------------------------------------------------- one config
def pmc_notes(cfg_name):
    """HBM traffic and VALU instruction counts of the dominant kernel from a builder-side `rocprofv3 --pmc` run of this same
    command (tools/pmc_fast.sh, tools/profile_gpu.sh; copied to profiles/): NOT measured in this process."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(cfg_name, {})
    except (OSError, ValueError):
        return {}
    # counters are quoted only for the library build they were taken on: tools/pmc_collect.py stores the hash of the library's
    # sources; a record without one, or with another, is older than the kernels and is refused
    from adcraft_amd import build as hip_build
    have, want = rec.get("library_source_hash"), hip_build.source_hash()
    if have != want:
        return {"stale": f"profiles/pmc_traffic.json[{cfg_name}] was collected on library sources {have}, this build is {want}: "
                         "PMC figures withheld (re-run tools/pmc_collect.py)"}
    return rec


def run_config(cfg_name, args, rank, world, local_rank, steps, warmup, with_cpu_baseline, collective_alone=False):
    from adcraft_amd import _ffi, synthetic, distributed as D
    from adcraft_amd.engine import StepEngine

    N, K, mean_volume, cvr, no_vol_prob, drift = synthetic.CONFIGS[cfg_name]
    planes = synthetic.implicit_keyword_planes(N, K, seed=1729 + rank, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
    n_dev = max(_ffi.device_count(), 1)
    eng = StepEngine(N, K, device_id=local_rank % n_dev, seed=1729, env_id_base=rank * N, max_days=MAX_DAYS,
                     loss_threshold=1.0e12, drift_enabled=drift, auto_reset=True)
    eng.set_all_params(planes)
    eng.reset()
    eng.sample_actions(0.30, 1.00, args.budget)  # actions resident in HBM before the timed region
    eng.metrics_enable(True)
    red = D.MetricReducer(eng, rank, world)      # world > 1: the engine's RCCL communicator (collective bring-up)
    collective_error = None
    if collective_alone and world == 1:
        # one GPU: the same RCCL communicator with a single rank, so that the one collective of the path runs - and is timed -
        # exactly where the N > 1 run has it (once per 60-step episode, inside the timed region)
        try:
            eng.comm_init(eng.comm_unique_id(), 0, 1)
        except Exception as exc:                 # no usable librccl on this box: say so in the line, keep measuring the steps
            collective_error = f"{type(exc).__name__}: {exc}"
    with_collective = world > 1 or (collective_alone and collective_error is None)
    # stationary keywords: the ideal (max expected) profit per keyword is constant over the episode
    ideal_nk = eng.ideal_profit(2048)            # (with drift on this is the episode-start value)
    ideal_k, ideal_pos_k = ideal_nk.sum(axis=0), np.where(ideal_nk <= 0, 1.0, ideal_nk).sum(axis=0)

    def barrier():
        eng.synchronize()
        red.barrier()

    for _ in range(warmup):
        eng.step_device()
    red.metric_sums(ideal_k * warmup, ideal_pos_k * warmup)       # also brings the collective path up outside the timed region
    barrier()
    # (the line also carries what the contract's flags alone measure - K steps right behind the W warm-up steps, before the
    #  clock spin below - so that the driver-shaped figure stays comparable with rounds that had no spin)
    t_u = time.perf_counter()
    for _ in range(steps):
        eng.step_device()
    barrier()
    unspun_ms = (time.perf_counter() - t_u) / steps * 1e3
    # a short timed region starts on clocks that are still ramping (a 20-step region measured 6 % slower than a 200-step one): more
    # UNTIMED steps until the device has been busy for args.spin_seconds; the driver's --warmup steps above stay what they are
    spin_steps, t_spin = 0, time.perf_counter()
    while time.perf_counter() - t_spin < args.spin_seconds:
        for _ in range(16):
            eng.step_device()
        eng.synchronize()
        spin_steps += 16
    barrier()
    # where the engine runs the step as env groups on several streams, the same K steps as ONE group (the schedule the kernel
    # times below are taken on): the figure the full-batch kernel's duration is to be read against
    eng.set_env_groups(1)
    barrier()
    t_g = time.perf_counter()
    for _ in range(steps):
        eng.step_device()
    barrier()
    one_group_ms = (time.perf_counter() - t_g) / steps * 1e3
    eng.set_env_groups(int(os.environ.get("ADCRAFT_STREAM_GROUPS", "0")))      # (back to the engine's choice, or the environment's)
    # kernel times: a separate, UNTIMED pass of the same length with HIP events around every step (four event records per step
    # keep a step's small tail kernels from overlapping the next step's launch: ~16 us of every step) ...
    eng.profile_enable(True, every=1)
    eng.profile_read()
    for _ in range(steps):
        eng.step_device()
    barrier()
    kernel_ms, launches = eng.profile_read()
    eng.profile_enable(False)
    # ... and the timed region itself records no event at all
    eng.metrics_reset()
    records_before = eng.profile_records()
    eng.comm_stats(reset=True)
    barrier()
    collective_host_s = 0.0
    eng.region_begin()                           # ONE event pair around the whole region: its GPU time, next to the host clock's
    t0 = time.perf_counter()
    for s in range(steps):
        eng.step_device()
        if with_collective and (s + 1) % MAX_DAYS == 0:
            tc = time.perf_counter()
            red.metric_sums(ideal_k * (s + 1), ideal_pos_k * (s + 1))     # the single collective of the path, once per episode
            collective_host_s += time.perf_counter() - tc
    barrier()
    elapsed = time.perf_counter() - t0
    groups = eng.env_groups()                    # how the engine scheduled the timed steps (env groups on their own streams)
    region_gpu_ms = eng.region_end()
    records_in_timed_region = eng.profile_records() - records_before
    coll_calls, coll_ms_local, coll_ms_allreduce = eng.comm_stats(reset=True)
    profit_c, ideal, ideal_pos, sc = red.metric_sums(ideal_k * steps, ideal_pos_k * steps)
    own_elapsed = elapsed
    elapsed = float(red.allreduce([elapsed], op="max")[0])          # the slowest rank's time
    per_rank = red.allreduce(np.eye(world)[rank] * own_elapsed)     # every rank's own time, next to the slowest
    _, ranks_in_comm = eng.comm_info()
    n_gpus = ranks_in_comm if red.backend == "rccl" else world

    units = float(world) * N * K * steps
    b_alg = BYTES_PER_U[drift] * N * K + BYTES_PER_ENV * N            # algorithmic bytes per launch (one GPU)
    dom = int(np.argmax(kernel_ms))              # the dominant kernel of the step
    step_kernel = eng.step_kernel_name()         # (which of the keyword-parallel kernels the engine chose for this workload)
    k_ms = kernel_ms[dom] / max(launches, 1)
    achieved = b_alg / (k_ms * 1e-3) / 1e9 if launches else None
    notes = pmc_notes(cfg_name) if dom == 0 else {}
    res = {
        "value": units / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps,
        "env_steps_per_s": float(world) * N * steps / elapsed, "n_gpus": n_gpus,
        "workload": f"{cfg_name}: {N} envs x {K} keywords per GPU, IMPLICIT keywords, mean_volume {mean_volume}, cvr {cvr}, "
                    f"no_vol_prob {no_vol_prob}, drift {drift}, budget {'non-binding' if args.budget >= 1e8 else args.budget}, "
                    f"{MAX_DAYS}-step episodes with auto-reset",
        "envs_per_gpu": N, "keywords": K,
        "clock_spin_steps": spin_steps,
        "ms_per_step_before_clock_spin": unspun_ms,
        "env_groups": {"timed_region": groups, "profiled_pass": 1, "ms_per_step_as_one_group": one_group_ms,
                       "note": (f"the timed steps (all but the first, which follows a barrier) ran as {groups} contiguous env groups of {N // groups} envs, each on its own HIP stream "
                                "(adc_engine_env_groups): one group's small kernels run under another's keyword-parallel pass and consecutive "
                                "steps of different groups overlap, so ms_per_step can be BELOW roofline.kernel_ms - that is the duration "
                                "of the kernel launched over all envs at once, as the profiled pass (always one group) launches it. "
                                "ADCRAFT_STREAM_GROUPS=1 runs the timed region the same way") if groups > 1 else
                               "one launch per kernel over all envs, in the timed region as in the profiled pass"},
        "timed_region": {"host_ms_per_step": own_elapsed / steps * 1e3, "gpu_ms_per_step": region_gpu_ms / steps,
                         "method": "host clock between two barriers (the contract's figure) | one HIP event pair on the engine's stream "
                                   "around the same region"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                     "traffic": notes.get("hbm_bytes_per_launch"),
                     "traffic_source": (notes.get("source_note") if notes.get("hbm_bytes_per_launch") else notes.get("stale")),
                     "kernel": step_kernel if dom == 0 else KERNEL_NAMES[dom], "kernel_ms": k_ms, "launches": int(launches), "launches_in_timed_region": steps if groups == 1 else 1 + (steps - 1) * groups,
                     "envs_per_launch": {"profiled_pass": N, "timed_region": N // groups},
                     "event_records_in_timed_region": int(records_in_timed_region),
                     "kernel_ms_method": f"HIP events on the engine's stream around every step of a separate, untimed pass of {steps} steps "
                                         "run between the warm-up and the timed region (same process, same state; the timed region itself "
                                         "records no event); rocprofv3 --kernel-trace --stats of the same command: profiles/",
                     "all_kernels_ms": {n: m / max(launches, 1) for n, m in zip((step_kernel,) + KERNEL_NAMES[1:], kernel_ms)},
                     "algorithmic_bytes_per_launch": b_alg},
    }
    valu = notes.get("valu")
    if valu and launches:
        # the bound that actually binds on dense keyword sets: VALU issue.  Instruction count: builder-side PMC (see `source`);
        # time: live, this run.  peak = one wave64 instruction per 2 cycles per SIMD-32 at 2.4 GHz - reachable only by the
        # full-rate opcodes (profiles/r02_issue_rates.md: compares, converts, min/max, integer multiplies cost 4 cycles)
        lane_ops = valu["wave_instructions_per_launch"] * 64.0 / (k_ms * 1e-3)
        res["roofline_valu"] = {"bound": "valu", "achieved": lane_ops, "peak": VALU_PEAK_LANE_OPS, "unit": "lane-instructions/s",
                                "frac": lane_ops / VALU_PEAK_LANE_OPS,
                                "peak_4cycle": VALU_PEAK_4CYCLE_LANE_OPS, "frac_4cycle": lane_ops / VALU_PEAK_4CYCLE_LANE_OPS,
                                "peaks": "peak: one wave64 instruction per 2 cycles per SIMD (packed / pure add-fma streams only); peak_4cycle: "
                                         "per 4 cycles - what compares, converts, min/max, integer multiplies and selects cost, i.e. the rate "
                                         "this kernel's instruction mix can reach (profiles/r02_issue_rates.md)",
                                "kernel": step_kernel, "kernel_ms": k_ms,
                                "valu_wave_instructions_per_launch": valu["wave_instructions_per_launch"],
                                "valu_lane_instructions_per_auction": valu.get("valu_lane_instructions_per_auction"),
                                "source": "instruction count: " + str(notes.get("source_note")) + "; kernel time: HIP events in this run"}
    elif notes.get("stale"):
        res["roofline_valu"] = {"bound": "valu", "achieved": None, "frac": None, "note": notes["stale"]}
    if world > 1:
        res["ms_per_step_by_rank"] = [float(x) / steps * 1e3 for x in per_rank]
    if with_collective or collective_error:
        n_eff = max(coll_calls, 1)
        res["collective"] = {
            "what": "adc_engine_metrics_allreduce: this rank's column sums of the per-(env, keyword) profit accumulators (k_metric_columns, "
                    f"k_metric_reduce, k_metric_vector), then ncclAllReduce(sum) of 3K + 8 = {3 * K + 8} doubles on the engine's stream",
            "ranks": n_gpus, "calls_in_timed_region": int(coll_calls), "every_steps": MAX_DAYS,
            "ms_per_call": {"own_reduction_kernels": coll_ms_local / n_eff, "allreduce": coll_ms_allreduce / n_eff,
                            "host_call_incl_waiting_for_the_steps_before_it": collective_host_s / n_eff * 1e3},
            "method": "three HIP events per call on the engine's stream (adc_engine_comm_stats); the host figure includes draining the "
                      "steps already enqueued",
        }
        if collective_error:
            res["collective"] = {"error": collective_error}
    akncp_ncp = D.episode_metrics(profit_c, ideal, ideal_pos, sc)
    res["episode_metric"] = {"AKNCP": akncp_ncp["AKNCP"], "NCP": akncp_ncp["NCP"], "profit_dollars": akncp_ncp["profit"],
                             "env_steps": akncp_ncp["env_steps"], "episodes": akncp_ncp["episodes"],
                             "note": "synthetic uniform bids, not a trained agent; envs pooled by keyword index"}
    if with_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg_name, planes, K, args.cpu_seconds)
    red.close()
    if collective_alone and world == 1 and collective_error is None:
        eng.comm_destroy()
    eng.close()
    return res


# ---------------------------------------------------------------------------------------------------------------- rank / launcher
def host_step(local_rank, steps=120):
    """what a numpy policy sees: BiddingSimulationVectorEnv.step at 4096 x 256, actions from host arrays, observations into
    host arrays (the reference's gymnasium call, SURVEY 8 row b).  PCIe-bound: 4 B per keyword up, 20 B down
    (14 B with uint16 counts)."""
    from adcraft_amd import gymnasium_kw_utils as utils
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv
    N, K = 4096, 256
    act = {"keyword_bids": np.full((N, K), 0.8, np.float32), "budget": np.full(N, 1e6, np.float32)}
    res = {}
    for key, opts in (("ms", {}), ("ms_u16_counts", {"compact_counts": True})):
        vec = BiddingSimulationVectorEnv(N, keyword_config=utils.experiment_keyword_config(128, 0.8), num_keywords=K, budget=1e6,
                                         param_sampler="device", device_id=local_rank, loss_threshold=1e12, **opts)   # 60-day episodes, autoreset
        vec.reset(seed=1)
        for _ in range(5):
            vec.step(act)
        per_step = np.empty(steps)
        for i in range(steps):
            t0 = time.perf_counter()
            vec.step(act)
            per_step[i] = time.perf_counter() - t0
        res[key] = round(float(np.median(per_step)) * 1e3, 4)
        res[key + "_mean_incl_episode_ends"] = round(float(per_step.mean()) * 1e3, 4)   # every 60th step copies the final observations out
        vec.close()
    res["value"] = round(N * K / (res["ms"] * 1e-3), 1)
    res.update(unit="keyword-steps/s", workload="cfg2 through BiddingSimulationVectorEnv.step (dict actions in, dict observations out)",
               engines_on_device=4, steps=steps)
    return res


def rehearse_rank(args, rank, world):
    """--rehearse: everything around the GPU work, without a GPU (tests/test_distributed_cpu.py): TWO bring-ups back to back,
    as the real N > 1 run has them (cfg4's engine, then cfg5's) - each an id hand-over and a reduction - the slowest-rank
    time, the line"""
    from adcraft_amd import comm, synthetic
    if args.fail_rank == rank:
        sys.stderr.write(f"bench.py: rank {rank} fails on request (--fail-rank)\n")
        raise SystemExit(3)
    seen, ids = [], []
    for bring_up in range(2):
        uid = comm.exchange_bytes(rank, world, lambda: bytes((bring_up + i) % 256 for i in range(128)), tag=f"id{comm.next_bring_up()}")
        red = comm.FileReducer(rank, world)
        seen.append(red.allreduce(np.eye(world)[rank] * (bring_up + 1)))
        elapsed = float(red.allreduce([0.001 * (rank + 1) * args.steps], op="max")[0])
        red.close()
        ids.append(uid)
    if rank == 0:
        N, K = synthetic.CONFIGS["cfg4"][:2]
        print(json.dumps({"metric": "env-steps/sec (envs×keywords auctions/s)", "value": None, "unit": "keyword-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "none (rehearsal)",
                          "config": {"workload": f"cfg4: {N} envs x {K} keywords per GPU (not run: --rehearse)"},
                          "rehearsal": {"ranks_seen": [int(i) for i in np.nonzero(seen[0])[0]], "id_bytes": len(ids[0]), "bring_ups": len(ids),
                                        "ids_distinct": ids[0] != ids[1], "second_reduction": [float(x) for x in seen[1]]}}), flush=True)


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse:
        return rehearse_rank(args, rank, world)
    # stdout carries ONE JSON line: whatever a library prints there (librccl's version banner at communicator bring-up) goes to stderr
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    main_cfg = args.config or ("cfg2" if world == 1 else "cfg4")
    also_cfgs = [] if args.no_also or args.config else (["cfg3", "cfg4", "cfg5"] if world == 1 else ["cfg5"])
    main = run_config(main_cfg, args, rank, world, local_rank, args.steps, args.warmup, with_cpu_baseline=(world == 1 and rank == 0 and not args.no_cpu_baseline),
                      collective_alone=(args.config == "cfg4" and not args.no_collective_alone))
    also = {}
    for c in also_cfgs:
        # (cfg4 at N = 1: the N > 1 workload's per-GPU shard - with its collective, on a one-rank communicator, so that its cost is on record)
        r = run_config(c, args, rank, world, local_rank, max(60 if c == "cfg4" else 50, min(args.steps, 100)), max(10, min(args.warmup, 20)),
                       with_cpu_baseline=False, collective_alone=(c == "cfg4" and not args.no_collective_alone))
        also[c] = {k: r[k] for k in ("value", "ms_per_step", "steps", "workload", "roofline", "episode_metric", "collective", "timed_region",
                                     "clock_spin_steps", "ms_per_step_before_clock_spin", "ms_per_step_by_rank", "env_groups") if k in r}
        if "roofline_valu" in r:
            also[c]["roofline_valu"] = r["roofline_valu"]
    if rank == 0:
        line = {
            "metric": "env-steps/sec (envs×keywords auctions/s)",
            "value": main["value"], "unit": "keyword-steps/s",
            "n_gpus": main["n_gpus"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32 cents + f32 (Philox4x32-7 u32)", "data": "synthetic",
            "config": {"workload": main["workload"], "envs_per_gpu": main["envs_per_gpu"], "keywords": main["keywords"],
                       "parallelism": f"env-sharded x{main['n_gpus']}",
                       "collective": ("none on the step path; RCCL all-reduce of the episode-metric vector once per episode "
                                      "(adc_engine_metrics_allreduce)") if world > 1 else "none (single GPU)",
                       # N = 1 measures BASELINE configs[1] (cfg2); N > 1 measures the 8-GPU config's per-GPU shard (cfg4):
                       # the one-GPU point of that weak-scaling curve is also.cfg4.value of the N = 1 line
                       "weak_scaling_reference": "also.cfg4.value of the N=1 line (same per-GPU shard as the N>1 runs)"},
            "env_steps_per_s": main["env_steps_per_s"],
            "roofline": main["roofline"],
        }
        for k in ("timed_region", "clock_spin_steps", "ms_per_step_before_clock_spin", "env_groups", "ms_per_step_by_rank", "collective", "roofline_valu", "episode_metric", "cpu_baseline"):
            if k in main:
                line[k] = main[k]
        if also:
            line["also"] = also
        if world == 1 and not args.no_also and not args.config:
            line["host_step"] = host_step(local_rank)
        print(json.dumps(line), file=line_out, flush=True)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """start args.gpus fresh rank processes (this process never touches a GPU), relay rank 0's line; a rank that fails - or
    the job outliving --launch-timeout - ends every rank, and what the failing ranks wrote to stderr is relayed"""
    import glob
    import shutil
    import tempfile
    port = free_port()
    job = f"bench{os.getpid()}"
    logdir = tempfile.mkdtemp(prefix=f"adcraft_{job}_")
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ADCRAFT_JOB_ID=job)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(open(os.path.join(logdir, f"rank{r}.err"), "w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1], text=True))

    def end_job(reason, relay_killed=False):
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        codes = [p.returncode for p in procs]
        sys.stderr.write(f"bench.py: {reason}; rank exit codes {codes}\n")
        for r, f in enumerate(errs):
            f.seek(0)
            tail = f.read()[-2000:]
            # (-9: a rank this launcher ended itself - of interest only when the job hung: then what every rank last said is the evidence)
            if tail.strip() and (relay_killed or codes[r] not in (0, -9)):
                sys.stderr.write(f"---- rank {r} stderr (tail) ----\n{tail}\n")
        for leftover in glob.glob(os.path.join(tempfile.gettempdir(), f"adcraft_comm_{port}_{job}.*")):
            shutil.rmtree(leftover, ignore_errors=True) if os.path.isdir(leftover) else os.remove(leftover)
        raise SystemExit(1)

    # rank 0's line is a few KB: it fits the pipe, so polling the exit codes first cannot deadlock on it
    deadline = time.monotonic() + args.launch_timeout
    try:
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):          # a rank failed: the others would wait for it forever
                end_job("a rank failed")
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                end_job(f"no result after --launch-timeout {args.launch_timeout:.0f} s", relay_killed=True)
            time.sleep(0.05)
        sys.stdout.write(procs[0].stdout.read())
        sys.stdout.flush()
        for r, f in enumerate(errs):                             # warnings of healthy ranks are not lost either
            f.seek(0)
            text = f.read()
            if text.strip():
                sys.stderr.write(text if r == 0 else f"[rank {r}] {text}")
    finally:
        for f in errs:
            f.close()
        shutil.rmtree(logdir, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, help="measure only this config (cfg2, cfg3, cfg4, cfg5) instead of the contract's workload")
    ap.add_argument("--budget", type=float, default=1.0e9, help="per-env daily budget in dollars (default: non-binding)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary configs")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--spin-seconds", type=float, default=0.25, help="untimed steps for this long after the warm-up, so that a short timed "
                    "region does not start on ramping clocks (0: none)")
    ap.add_argument("--no-collective-alone", action="store_true", help="N = 1: do not bring up the one-rank RCCL communicator for cfg4")
    ap.add_argument("--rehearse", action="store_true", help="no GPU work: launch, id hand-over and reduction only (CPU test)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="seconds the self-started ranks of --gpus N may take before the job is ended")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    main()
