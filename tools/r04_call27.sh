set -e
mkdir -p gpurun_out/c27
timeout -k 10 200 python tools/sorted_gemm_bench.py > gpurun_out/c27/base.log 2>&1
for v in 1 2 4 7; do
LIDAR_HIP_SO=$PWD/lidardetection_amd/csrc/liblidar_hip_sc_p$v.so timeout -k 10 200 python tools/sorted_gemm_bench.py > gpurun_out/c27/p$v.log 2>&1
done
