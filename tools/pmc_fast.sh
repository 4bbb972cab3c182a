#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of k_step_implicit_fast for one config (two PMC passes), printed as a table.
# usage: tools/pmc_fast.sh <outdir> [bench args...]
export TMPDIR=/tmp
OUT=$1; shift
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -o pmc -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $OUT/log1.txt 2>&1 || { tail -5 $OUT/log1.txt; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o pmc -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $OUT/log2.txt 2>&1 || { tail -5 $OUT/log2.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step_implicit" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} {sum(v) / len(v):.5g}   (n={len(v)})")
PY
