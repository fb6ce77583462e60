set -e
mkdir -p gpurun_out/c28
for r in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/c28/base_$r.log 2>&1
LIDAR_HIP_SO=$PWD/lidardetection_amd/csrc/liblidar_hip_vxl_fill_nt.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/c28/nt_$r.log 2>&1
done
