#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + PMC passes of bench.py; raw output under gpurun_out/.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:---steps 60 --warmup 5 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1 || { tail -5 $OUT/bench_stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline ${ARGS#*--no-cpu-baseline} > $OUT/bench_fetch.log 2>&1 || { tail -5 $OUT/bench_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline ${ARGS#*--no-cpu-baseline} > $OUT/bench_write.log 2>&1 || { tail -5 $OUT/bench_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline ${ARGS#*--no-cpu-baseline} > $OUT/bench_sq.log 2>&1 || { tail -5 $OUT/bench_sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline ${ARGS#*--no-cpu-baseline} > $OUT/bench_sq2.log 2>&1 || { tail -5 $OUT/bench_sq2.log; }
find $OUT -name "*.csv" | head -30
python tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1; cat $OUT/summary.md
