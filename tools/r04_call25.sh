set -e
mkdir -p gpurun_out/c25
for r in 1 2; do
LIDAR_HIP_SO=$PWD/lidardetection_amd/csrc/liblidar_hip_w43_aux0.so timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/c25/bench_aux0_$r.log 2>&1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/c25/bench_aux2_$r.log 2>&1
done
