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
    if os.environ.get("ADCRAFT_ORACLE_SANITIZE") == "1":
        return build_sanitized()[2]
    if force or not os.path.exists(BATTERY) or os.path.getmtime(BATTERY) < os.path.getmtime(BATTERY_SRC):
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp", "-Wall", "-Wextra",
                               BATTERY_SRC, "-o", BATTERY, "-lm"])
    return BATTERY


def build(force=False):
    if os.environ.get("ADCRAFT_ORACLE_SANITIZE") == "1":
        return build_sanitized()[:2]
    build_battery(force)
    outs = []
    for name, extra in (("libadcraft_oracle.so", []), ("libadcraft_oracle_fma.so", ["-mfma", "-mavx2"])):
        out = os.path.join(HERE, name)
        if force or _stale(out):
            subprocess.check_call(COMMON + extra + [SRC, "-o", out, "-lm"])
        outs.append(out)
    return outs


# ---- sanitizer builds (host ASan/UBSan of the CPU restatement; CPU only - GPU ASan does not exist on this pool) ----
SAN_DIR = os.path.join(HERE, "_san")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
SHIMS_SRC = os.path.join(os.path.dirname(HERE), "adcraft_amd", "csrc", "adc_shims.cpp")


def build_sanitized():
    """oracle/_san/: the oracle (both variants), the stream battery and a HOST-ONLY g++ build of the product's scalar shims
    (adcraft_amd/csrc/adc_shims.cpp + adc_law.h, otherwise only ever compiled by hipcc), all with -fsanitize=address,undefined and
    -fno-sanitize-recover.  Loaded instead of the plain builds when ADCRAFT_ORACLE_SANITIZE=1 (python must then run with
    LD_PRELOAD=$(gcc -print-file-name=libasan.so): tools/run_sanitized.sh)."""
    os.makedirs(SAN_DIR, exist_ok=True)
    flags = [f for f in COMMON if f != "-O2"] + SAN_FLAGS
    outs = []
    for name, extra in (("libadcraft_oracle.so", []), ("libadcraft_oracle_fma.so", ["-mfma", "-mavx2"])):
        out = os.path.join(SAN_DIR, name)
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC):
            subprocess.check_call(flags + extra + [SRC, "-o", out, "-lm"])
        outs.append(out)
    bat = os.path.join(SAN_DIR, "libstream_battery.so")
    if not os.path.exists(bat) or os.path.getmtime(bat) < os.path.getmtime(BATTERY_SRC):
        subprocess.check_call(["gcc", "-std=gnu11", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp", "-Wall", "-Wextra"] + SAN_FLAGS +
                              [BATTERY_SRC, "-o", bat, "-lm"])
    outs.append(bat)
    outs.append(build_shims_host(sanitize=True))
    return outs


def build_shims_host(sanitize=False):
    """adc_shims.cpp as plain host C++ (g++, no HIP): the scalar `adcraft.rust` entry points and the word-interval / bracket checkers
    of adc_law.h, as a library of their own - proof that this file is host code, and the form the sanitizers can see"""
    out_dir = SAN_DIR if sanitize else os.path.join(HERE, "_host")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libadc_shims_host.so")
    deps = [SHIMS_SRC, os.path.join(os.path.dirname(SHIMS_SRC), "adc_law.h")]
    if not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra",
                               "-Wno-unused-function", "-Wno-unknown-pragmas"] + (SAN_FLAGS if sanitize else ["-O2"]) + [SHIMS_SRC, "-o", out, "-lm"])
    return out


if __name__ == "__main__":
    import sys
    for p in (build_sanitized() if "--sanitize" in sys.argv else build(force=True) + [build_shims_host()]):
        print("built", p)
