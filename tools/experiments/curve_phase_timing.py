#!/usr/bin/env python3
"""Developer tool (needs a -DADC_EXP_TIMING build with phase marks in k_curve_contenders - a temporary patch, see profiles/r05_curve_contenders.txt):
cycles per keyword of the kernel's passes at cfg2 size.  ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/timing.so python tools/experiments/curve_phase_timing.py"""
import ctypes as C
import sys
sys.path.insert(0, ".")
from adcraft_amd import _ffi, synthetic
from adcraft_amd.engine import StepEngine
N, K = 4096, 256
planes = synthetic.implicit_keyword_planes(N, K, seed=1729, mean_volume=128, cvr=0.8)
e = StepEngine(N, K, seed=1729, drift_enabled=True, max_days=60, loss_threshold=1e12, auto_reset=True)
e.set_all_params(planes); e.reset(); e.synchronize()
L = _ffi.lib()
L.adc_debug_read.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
L.adc_debug_read(out, 1)
e.bid_curves_build(2048)
e.synchronize()
L.adc_debug_read(out, 0)
v = [float(x) / (N * K) for x in list(out)[8:]]
for name, x in zip(("load lines", "hull (lane 0)", "intervals", "compaction + monotone ends", "write-out"), v[:5]):
    print(f"{name:30s} {x:9.0f} cycles per keyword")
print(f"lines {v[5]:.1f}  hull {v[6]:.1f}  contenders {v[7]:.1f} per keyword; total {sum(v[:5]):.0f} cycles")
e.close()
