#!/bin/bash
# builds lidardetection_amd/csrc/liblidar_hip_stamps.so = the library with -DVXL_STAMPS in voxelize.hip (tools/vx_phase_probe.py) and
# -DTK_STAMPS in topk.hip (tools/topk_phase_probe.py)
set -e
cd "$(dirname "$0")/../lidardetection_amd/csrc"
python build.py > /dev/null
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
objs=$(ls *.o | grep -v -e "^voxelize.o$" -e "^topk.o$")
for v in ""; do
  /opt/rocm/bin/hipcc $F -DVXL_STAMPS ${v:+-DVXL_EXP=$v} -c voxelize.hip -o /tmp/voxelize_stamps$v.o
  /opt/rocm/bin/hipcc $F -DTK_STAMPS -c topk.hip -o /tmp/topk_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblidar_hip_stamps$v.so $objs /tmp/voxelize_stamps$v.o /tmp/topk_stamps.o
  echo built liblidar_hip_stamps$v.so
done
