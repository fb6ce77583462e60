set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_vx2; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_pointpillar_path.py tests/test_gpu_roi_pool.py -m gpu -x -q -k "voxelize or points_in_boxes or boundary" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for c in uniform ring; do
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c --resident >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c --flush --iters 30 >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c --resident --flush --iters 30 >> $O/vx.log 2>&1
done
grep algo $O/vx.log
export LIDAR_HIP_SO=$PWD/lidardetection_amd/csrc/liblidar_hip_stamps.so
for a in "--flush" "--flush --resident"; do echo "== $a"; timeout -k 10 100 python tools/vx_phase_probe.py $a 2>&1 | tail -4; done
