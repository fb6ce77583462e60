#!/bin/bash
# builds lidardetection_amd/csrc/liblidar_hip_stamps.so = the library with -DVXL_STAMPS in voxelize.hip (tools/vx_phase_probe.py)
set -e
cd "$(dirname "$0")/../lidardetection_amd/csrc"
python build.py > /dev/null
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
objs=$(ls *.o | grep -v '^voxelize.o$')
for v in "" 1 2; do     # plain stamps, and the phase-A1 probes (VXL_EXP 1: loads only, 2: no loads)
  /opt/rocm/bin/hipcc $F -DVXL_STAMPS ${v:+-DVXL_EXP=$v} -c voxelize.hip -o /tmp/voxelize_stamps$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblidar_hip_stamps$v.so $objs /tmp/voxelize_stamps$v.o
  echo built liblidar_hip_stamps$v.so
done
