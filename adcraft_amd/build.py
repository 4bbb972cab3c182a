"""Build the HIP engine for gfx950 in-tree: adcraft_amd/lib/libadcraft_hip.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps every float32 rounding of
csrc/adc_law.h a single IEEE operation (the CPU oracle reproduces them bit for bit);
correctly rounded f32 divide/sqrt is hipcc's default and is requested explicitly.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libadcraft_hip.so")
SOURCES = ["adc_engine.hip", "adc_shims.cpp"]
HEADERS = ["adc_law.h", os.path.join(ROOT, "include", "adcraft_engine.h")] + sorted(
    os.path.join("parts", f) for f in os.listdir(os.path.join(SRC_DIR, "parts")) if f.endswith(".inc"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wall", "-Wno-unused-function"]


def _deps():
    out = [os.path.join(SRC_DIR, s) for s in SOURCES]
    out += [h if os.path.isabs(h) else os.path.join(SRC_DIR, h) for h in HEADERS]
    return [p for p in out if os.path.exists(p)]


def source_hash():
    """sha256 (16 hex digits) over the library's sources in a fixed order: what tools/pmc_collect.py stores next to its counters
    and bench.py compares, so that counters taken on other kernels are not quoted as this build's"""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(_deps()):
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False, extra=()):
    """Compile if the library is missing or older than a source.  Safe under several processes at once (one rank per GPU
    all import the package): the compile runs under an exclusive file lock into a temporary file that is renamed over
    the library, so a concurrent dlopen sees the old library or the new one, never a partial file."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and not stale():
        return LIB
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():          # another process built it while this one waited
                return LIB
            srcs = [os.path.join(SRC_DIR, s) for s in SOURCES if os.path.exists(os.path.join(SRC_DIR, s))]
            tmp = f"{LIB}.tmp{os.getpid()}"
            cmd = [HIPCC] + FLAGS + list(extra) + srcs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def build_timing_variant():
    """lib/variants/timing.so: the same library with in-kernel phase timers (-DADC_EXP_TIMING); select it with
    ADCRAFT_HIP_LIB=<path> (tools/exp_rows_timing.py).  Never loaded by default."""
    out_dir = os.path.join(LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "timing.so")
    srcs = [os.path.join(SRC_DIR, s) for s in SOURCES]
    subprocess.check_call([HIPCC] + FLAGS + ["-DADC_EXP_TIMING", "-Wno-unused-value"] + srcs + ["-o", out])
    return out


def build_variant(name, defines):
    """lib/variants/<name>.so: the library with extra -D switches (ablations, experiments); select it with ADCRAFT_HIP_LIB=<path>.
    Never loaded by default."""
    out_dir = os.path.join(LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    srcs = [os.path.join(SRC_DIR, s) for s in SOURCES]
    subprocess.check_call([HIPCC] + FLAGS + ["-Wno-unused-value", "-Wno-unused-variable"] + ["-D" + d for d in defines] + srcs + ["-o", out])
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python adcraft_amd/build.py --variant <name> <DEFINE[=value]> ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    elif "--timing" in sys.argv:
        print(build_timing_variant())
    else:
        print(build(force="--force" in sys.argv, verbose=True))
