set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_fused; mkdir -p $O; rm -f $O/*.log
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_pointpillar_path.py tests/test_gpu_bench_paths.py -m gpu -x -q -k "voxel or pointpillar_kitti_bs16" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for tw in 0 1; do
  export LIDAR_VXL_TWO_LAUNCH=$tw
  echo "two_launch=$tw" >> $O/vx.log
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --flush --iters 30 >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --resident --flush --iters 30 >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud ring --flush --iters 30 >> $O/vx.log 2>&1
done
grep -E "algo|two_launch" $O/vx.log
export LIDAR_VXL_TWO_LAUNCH=0
for v in ""; do
export LIDAR_HIP_SO=$R/lidardetection_amd/csrc/liblidar_hip_stamps$v.so
timeout -k 10 120 python tools/vx_phase_probe.py --flush --resident > $O/phase_res$v.log 2>&1
timeout -k 10 120 python tools/vx_phase_probe.py --flush > $O/phase_full$v.log 2>&1
echo "variant $v"
tail -4 $O/phase_res$v.log
tail -4 $O/phase_full$v.log
done
