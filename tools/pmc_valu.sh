#!/bin/bash
# quick VALU-instruction count of the dominant kernel (one PMC pass)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_valu; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "implicit_fast" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "%.4g" % (sum(v) / len(v)))
PY
