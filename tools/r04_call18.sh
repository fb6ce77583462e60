set -e
mkdir -p gpurun_out/c18
timeout -k 10 400 python -m pytest tests/test_gpu_wino.py -x -q -m gpu -k "f43" > gpurun_out/c18/f43_tests.log 2>&1
timeout -k 10 800 python tools/wino43_probe.py "0 0:W43_R=6 0:W43_B2=60 0:W43_B1=24,W43_B2=48 0:W43_B1=30,W43_B2=64 0:W43_R=12" 16,64,248,216 16,128,124,108 16,256,62,54 > gpurun_out/c18/probe.log 2>&1
