"""GPU box: the closed loop (agent + per-step ideal + step) as ONE engine against FOUR engines of a quarter of the envs each, every one
on its own stream - what group-local policy kernels could buy"""
import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
N, K, mean_volume, cvr, no_vol_prob, _ = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mean_volume, cvr=cvr, no_vol_prob=no_vol_prob)
for shards in (1, 4, 1, 4):
    n = N // shards
    engs = []
    for s in range(shards):
        e = StepEngine(n, K, seed=1729 + s, max_days=60, loss_threshold=1e12, drift_enabled=True, auto_reset=True)
        e.set_all_params(planes[:, s * n:(s + 1) * n])
        e.reset()
        e.bid_curves_build(2048)
        e.metrics_enable(True)
        e.agent_init(1.0, None)
        engs.append(e)
    def loop(with_ideal):
        for e in engs:
            e.agent_step(100000.0)
            if with_ideal:
                e.ideal_step(fetch=False)
            e.step_device()
    out = []
    for with_ideal in (True, False):
        for _ in range(30):
            loop(with_ideal)
        for e in engs:
            e.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            loop(with_ideal)
        for e in engs:
            e.synchronize()
        out.append(f"{'agent+ideal+step' if with_ideal else 'agent+step'} {(time.perf_counter() - t0) / 100 * 1e3:.4f}")
    print(f"{shards} engine(s): " + "  ".join(out), flush=True)
    for e in engs:
        e.close()
