set -e
mkdir -p gpurun_out/c31
timeout -k 10 600 python -m pytest tests/test_gpu_spconv.py -x -q -m gpu -k "two_row" > gpurun_out/c31/tests.log 2>&1
