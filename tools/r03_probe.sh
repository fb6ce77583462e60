set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_probe; mkdir -p $O
cd $R
export LIDAR_HIP_SO=$R/lidardetection_amd/csrc/liblidar_hip_stamps.so
timeout -k 10 120 python tools/vx_phase_probe.py --flush --resident > $O/phase_res.log 2>&1
timeout -k 10 120 python tools/vx_phase_probe.py --flush > $O/phase_full.log 2>&1
tail -9 $O/phase_res.log
tail -9 $O/phase_full.log
