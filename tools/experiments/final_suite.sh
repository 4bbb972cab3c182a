set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_cfg2.json 2> gpurun_out/final/bench_cfg2.err || exit 1
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --steps 120 --warmup 10 --no-cpu-baseline > gpurun_out/final/bench_$c.json 2>> gpurun_out/final/err.txt || exit 1; done
for b in 1000 10 1; do python bench.py --budget $b --steps 120 --warmup 10 --no-cpu-baseline > gpurun_out/final/bench_budget$b.json 2>> gpurun_out/final/err.txt || exit 1; done
bash tools/profile_gpu.sh r01_final3 > gpurun_out/final/prof.log 2>&1 || exit 1
python tools/measure_host_step.py > gpurun_out/final/host_step.txt 2>&1 || exit 1
python tools/measure_vector_env.py > gpurun_out/final/vector_env.txt 2>&1 || exit 1
python tools/measure_small_env.py > gpurun_out/final/small_env.txt 2>&1 || exit 1
echo done
