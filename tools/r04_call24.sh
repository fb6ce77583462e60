set -e
mkdir -p gpurun_out/c24
timeout -k 10 300 python bench.py --stages --no-cpu-baseline > gpurun_out/c24/bench_nt.log 2>&1
