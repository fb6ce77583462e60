set -e
mkdir -p gpurun_out/c29
timeout -k 10 600 python -m pytest tests/test_gpu_pointpillar_path.py tests/test_gpu_bench_paths.py -x -q -m gpu -k "voxel or vox or feeder or timed" > gpurun_out/c29/tests.log 2>&1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/c29/share16.log 2>&1
for v in 12 10 8; do
LIDAR_HIP_SO=$PWD/lidardetection_amd/csrc/liblidar_hip_vxl_share$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra > gpurun_out/c29/share$v.log 2>&1
done
