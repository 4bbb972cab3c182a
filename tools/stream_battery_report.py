#!/usr/bin/env python3
"""Philox4x32 with 7 rounds (the stream) against 10 rounds (Random123's default) on the production counter layout:
the battery of oracle/stream_battery.c at 6e8 words per round count and two seeds.  Usage: python tools/stream_battery_report.py > profiles/r03_stream_battery.txt"""
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.test_stream_quality import AXES, STATS, run  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
print(f"stream battery (oracle/stream_battery.c): z-scores, N(0,1) under the null; {calls} Philox calls per axis")
for rounds in (7, 10):
    for seed in (20261004, 77):
        words, stats, pairs, derived = run(rounds, calls, seed)
        print(f"\nPhilox4x32-{rounds}, seed {seed}, {words:.3g} words")
        print(f"{'axis':28s} " + " ".join(f"{s:>19s}" for s in STATS))
        for a, name in enumerate(AXES):
            print(f"{name:28s} " + " ".join(f"{stats[a, s]:19.2f}" for s in range(8)))
        print("positions of one call (0,1) (1,2) (2,3) (0,3): chi2 top / bottom bytes  " + "  ".join(f"{pairs[p, 0]:.2f}/{pairs[p, 1]:.2f}" for p in range(4)))
        print(f"derived pair: click vs competitor uniform independence z {derived[0]:.2f}; click frequency z {derived[1]:.2f}; competitor uniform top-12-bit uniformity z {derived[2]:.2f}")
        allz = np.concatenate([stats.ravel(), pairs.ravel(), derived])
        print(f"all {allz.size} statistics: max |z| {np.abs(allz).max():.2f}, mean {allz.mean():.2f}, std {allz.std():.2f}")
for rounds in (2, 3, 4, 5):
    _, stats, pairs, derived = run(rounds, 1_000_000, 5)
    print(f"\n(for scale) Philox4x32-{rounds} at 1e6 calls per axis: max |z| {max(np.abs(stats).max(), np.abs(pairs).max(), np.abs(derived).max()):.1f}")
