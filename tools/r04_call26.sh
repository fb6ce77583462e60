set -e
mkdir -p gpurun_out/c26
timeout -k 10 600 python -m pytest tests/test_gpu_deconv.py tests/test_gpu_pointpillar_path.py -x -q -m gpu > gpurun_out/c26/tests.log 2>&1
timeout -k 10 200 python tools/deconv_bench.py > gpurun_out/c26/dc_new.log 2>&1
timeout -k 10 300 python bench.py --stages --no-cpu-baseline > gpurun_out/c26/bench.log 2>&1
