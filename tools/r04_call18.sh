set -e
mkdir -p gpurun_out/c18
timeout -k 10 400 python -m pytest tests/test_gpu_wino.py -x -q -m gpu -k "f43" > gpurun_out/c18/f43_tests.log 2>&1
timeout -k 10 800 python tools/wino43_probe.py "0 1024 1280 256" 16,64,248,216 > gpurun_out/c18/probe.log 2>&1
