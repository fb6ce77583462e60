set -e
mkdir -p gpurun_out/c16
timeout -k 10 400 python -m pytest tests/test_gpu_wino.py -x -q -m gpu -s -k "f43" > gpurun_out/c16/f43_tests.log 2>&1
WINO_BENCH_ONLY=1 timeout -k 10 300 python tools/wino_bench.py > gpurun_out/c16/bench.log 2>&1
