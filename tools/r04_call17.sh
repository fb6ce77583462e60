set -e
mkdir -p gpurun_out/c17
timeout -k 10 800 python tools/wino43_probe.py "0 8 4 2 1 16 12 31" 16,64,248,216 16,128,124,108 16,256,62,54 > gpurun_out/c17/probe.log 2>&1
