set -e
mkdir -p gpurun_out/c24
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_deconv.py -x -q -m gpu > gpurun_out/c24/tests.log 2>&1
timeout -k 10 300 python bench.py --stages --no-cpu-baseline > gpurun_out/c24/bench_nt2.log 2>&1
