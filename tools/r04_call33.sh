set -e
mkdir -p gpurun_out/c33
LIDAR_WINO_F43=0 timeout -k 10 900 python -m pytest tests/test_gpu_pointpillar_path.py tests/test_gpu_bench_paths.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/c33/tests_f23.log 2>&1
LIDAR_WINO_F43=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/c33/bench_f23.log 2>&1
