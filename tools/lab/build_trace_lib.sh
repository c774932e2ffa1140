#!/bin/bash
# lab: libpcs_hip with the bucket reduce's phase stamps and the workspace address export (-DPCS_BKT_TRACE), linked from the objects
# of the last `make -C pointcloud_stitching_amd/csrc`; load it with PCS_LIB_PATH=$PWD/pointcloud_stitching_amd/lib/lab/libpcs_hip_trace.so
set -e
cd "$(dirname "$0")/../../pointcloud_stitching_amd/csrc"
mkdir -p ../lib/lab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-parameter -DPCS_BKT_TRACE=1 -c pcs_voxel.hip -o /tmp/pcs_voxel_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/lab/libpcs_hip_trace.so pcs_kernels.o pcs_kernels_voxel.o /tmp/pcs_voxel_trace.o pcs_capi.o pcs_capi_voxel.o
ls -la ../lib/lab/libpcs_hip_trace.so
