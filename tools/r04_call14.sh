set -e
mkdir -p gpurun_out/c14
timeout -k 10 300 python -m pytest tests/test_gpu_pointpillar_path.py -x -q -m gpu -k "nms" > gpurun_out/c14/nms_tests.log 2>&1
timeout -k 10 200 python tools/nms_keep_pos.py > gpurun_out/c14/keep_pos.log 2>&1
timeout -k 10 300 python bench.py --stages > gpurun_out/c14/bench.log 2>&1
