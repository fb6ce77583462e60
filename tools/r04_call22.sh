set -e
mkdir -p gpurun_out/c22
LIDAR_BEV_SPLIT=4 LIDAR_BEV_SPLIT_MIN=4 timeout -k 10 300 python bench.py --stages --no-cpu-baseline > gpurun_out/c22/bench_split4.log 2>&1
LIDAR_BEV_SPLIT=2 LIDAR_BEV_SPLIT_MIN=4 timeout -k 10 300 python bench.py --stages --no-cpu-baseline > gpurun_out/c22/bench_split2min4.log 2>&1
