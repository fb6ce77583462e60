set -e
mkdir -p gpurun_out/c19
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/c19/gpu_tests.log 2>&1
timeout -k 10 400 python bench.py --stages > gpurun_out/c19/bench.log 2>&1
