set -e
mkdir -p gpurun_out/c32
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/c32/gpu_tests.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/c32/bench.log 2>&1
