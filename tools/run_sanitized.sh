#!/bin/bash
# CPU-only sanitizer run (GPU AddressSanitizer does not exist on this pool): the oracle, the stream battery and a
# host-only g++ build of the product's scalar shims (adc_shims.cpp + adc_law.h) are built with -fsanitize=address,undefined
# -fno-sanitize-recover (python oracle/build.py --sanitize -> oracle/_san/) and the whole `-m "not gpu"` suite runs on them.
# Usage: bash tools/run_sanitized.sh [extra pytest arguments]
set -euo pipefail
cd "$(dirname "$0")/.."
python oracle/build.py --sanitize
export ADCRAFT_ORACLE_SANITIZE=1
# (leak detection off: the interpreter and the HIP runtime the product library pulls in never free everything at exit)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD="$(gcc -print-file-name=libasan.so)" python -m pytest tests -m "not gpu" -x -q "$@"
