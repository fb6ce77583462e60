# end-of-round evidence run (one gpurun call): full GPU suite, the driver's bench line, and the kernel trace of the same bench
set -e
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1
timeout -k 10 400 python bench.py --stages > gpurun_out/final/bench.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -o pp -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --roofline-launches 5 > $R/gpurun_out/final/prof.log 2>&1
cd $R
LIDAR_BEV_SPLIT=1 python tools/ktrace_last.py gpurun_out/final/prof 100 vxl_keybin > gpurun_out/final/step_timeline.txt 2>&1 || true
