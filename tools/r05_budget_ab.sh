#!/bin/bash
# Runs ON THE GPU BOX: cfg2's step time at one budget for the product build and variant builds, twice, + the row kernel's VALU count
# usage: tools/r05_budget_ab.sh <budget> <variant> ...
export TMPDIR=/tmp
b=$1; shift
for rep in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then unset ADCRAFT_HIP_LIB ADCRAFT_ALLOW_STALE_LIB; else export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1; fi
  timeout -k 10 120 python3 - $b <<'PY' 2>&1 | tail -1 | sed "s/^/$v: /"
import sys, time
sys.path.insert(0, ".")
from adcraft_amd import synthetic
from adcraft_amd.engine import StepEngine
budget = float(sys.argv[1])
N, K, mv, cvr, nv, drift = synthetic.CONFIGS["cfg2"]
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=mv, cvr=cvr, no_vol_prob=nv)
eng = StepEngine(N, K, seed=1729, max_days=1 << 30, loss_threshold=1e15)
eng.set_all_params(planes); eng.reset(); eng.sample_actions(0.30, 1.00, budget)
for _ in range(3):
    for _ in range(4): eng.step_device()
    eng.synchronize()
t0 = time.perf_counter()
for _ in range(60): eng.step_device()
eng.synchronize()
print(f"budget {budget:g}: {(time.perf_counter() - t0) / 60 * 1e3:.4f} ms/step")
PY
done
done
