set -e
mkdir -p gpurun_out/c23
timeout -k 10 600 python -m pytest tests/test_gpu_spconv.py -x -q -m gpu > gpurun_out/c23/spconv_tests.log 2>&1
timeout -k 10 300 python tools/sorted_gemm_bench.py > gpurun_out/c23/gemm_layers.log 2>&1
LIDAR_SPCONV_RT2=0 timeout -k 10 300 python tools/sorted_gemm_bench.py > gpurun_out/c23/gemm_layers_rt1.log 2>&1
