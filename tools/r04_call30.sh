set -e
mkdir -p gpurun_out/c30
timeout -k 10 400 python -m pytest tests/test_gpu_wino.py -x -q -m gpu -k "random" > gpurun_out/c30/tests.log 2>&1
