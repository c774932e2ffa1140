#!/bin/bash
# lab: libpcs_hip variants whose voxel readers (pcs_kernels.hip as its second translation unit) are built with extra -D flags,
# linked from the other objects of the last `make -C pointcloud_stitching_amd/csrc`.
#   tools/lab/build_vox_variants.sh name1 "-DFLAG=..." [name2 "-D..."] ...   ->  pointcloud_stitching_amd/lib/lab/libpcs_hip_<name>.so
# load with PCS_LIB_PATH=$PWD/pointcloud_stitching_amd/lib/lab/libpcs_hip_<name>.so
set -e
cd "$(dirname "$0")/../../pointcloud_stitching_amd/csrc"
make -s >/dev/null
mkdir -p ../lib/lab
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
      -Wno-unused-parameter -mllvm -amdgpu-kernarg-preload-count=16 -DPCS_TU_VOXEL=1 -Wno-unused -Wno-unneeded-internal-declaration \
      $defs -c pcs_kernels.hip -o /tmp/pcs_kernels_voxel_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/lab/libpcs_hip_$name.so pcs_kernels.o /tmp/pcs_kernels_voxel_$name.o pcs_voxel.o pcs_capi.o pcs_capi_voxel.o
  ls -la ../lib/lab/libpcs_hip_$name.so
done
