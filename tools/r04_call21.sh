set -e
mkdir -p gpurun_out/c21
timeout -k 10 400 python -m pytest tests/test_gpu_wino.py -x -q -m gpu > gpurun_out/c21/wino_tests.log 2>&1
LIDAR_BEV_SPLIT=1 timeout -k 10 300 python bench.py --stages --no-cpu-baseline --no-extra > gpurun_out/c21/bench_split1.log 2>&1
LIDAR_BEV_SPLIT=2 timeout -k 10 300 python bench.py --stages --no-cpu-baseline --no-extra > gpurun_out/c21/bench_split2.log 2>&1
