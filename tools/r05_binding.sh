#!/bin/bash
# Runs ON THE GPU BOX: the binding-budget iteration loop - the budget-exact parity tests, then cfg2's step time at budgets 1000 / 10 / 1 (and the
# other shapes of tools/exp_binding.py) for the product build and any variant builds given
export TMPDIR=/tmp
OUT=gpurun_out/r05b; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py tests/test_gpu_device_resident.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
tail -4 $OUT/tests.log
[ $rc = 0 ] || exit $rc
for v in product "$@"; do
  if [ $v = product ]; then unset ADCRAFT_HIP_LIB ADCRAFT_ALLOW_STALE_LIB; else export ADCRAFT_HIP_LIB=adcraft_amd/lib/variants/$v.so ADCRAFT_ALLOW_STALE_LIB=1; fi
  echo "== $v" | tee -a $OUT/binding.txt
  timeout -k 10 300 python3 tools/exp_binding.py 2>&1 | tee -a $OUT/binding.txt || exit 1
done
