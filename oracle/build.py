"""Compile the CPU oracle (test infrastructure; see adcraft_oracle.c header).

Two shared objects are produced so the same snapshot runs on any x86-64 host:
  libadcraft_oracle_fma.so   -mfma -mavx2 (fmaf inlined; used when the CPU has FMA)
  libadcraft_oracle.so       baseline x86-64 (fmaf through libm; bit-identical results)
-ffp-contract=off is what makes the float32 transforms reproducible: every rounding in
the source is one IEEE operation and the compiler may not fuse or reorder them.

There is no reference build here (oracle/_ref is not produced): the reference's native path
is a Rust/pyo3 crate with un-vendored dependencies and no Rust toolchain exists in this
image, so it is unbuildable; see DESIGN.md.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "adcraft_oracle.c")
COMMON = ["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off",
          "-fno-fast-math", "-fopenmp", "-Wall", "-Wextra", "-Wno-unused-parameter"]


def _stale(out):
    return (not os.path.exists(out)) or os.path.getmtime(out) < os.path.getmtime(SRC)


BATTERY_SRC = os.path.join(HERE, "stream_battery.c")
BATTERY = os.path.join(HERE, "libstream_battery.so")


def build_battery(force=False):
    """the stream-quality battery (stream_battery.c): plain integer code, one build"""
    if force or not os.path.exists(BATTERY) or os.path.getmtime(BATTERY) < os.path.getmtime(BATTERY_SRC):
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp", "-Wall", "-Wextra",
                               BATTERY_SRC, "-o", BATTERY, "-lm"])
    return BATTERY


def build(force=False):
    build_battery(force)
    outs = []
    for name, extra in (("libadcraft_oracle.so", []), ("libadcraft_oracle_fma.so", ["-mfma", "-mavx2"])):
        out = os.path.join(HERE, name)
        if force or _stale(out):
            subprocess.check_call(COMMON + extra + [SRC, "-o", out, "-lm"])
        outs.append(out)
    return outs


if __name__ == "__main__":
    for p in build(force=True):
        print("built", p)
