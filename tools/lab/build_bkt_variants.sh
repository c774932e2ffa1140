#!/bin/bash
# lab: libpcs_hip variants whose voxel tail (pcs_voxel.hip) is built with extra -D flags; the other objects from the last make.
#   tools/lab/build_bkt_variants.sh name1 "-DFLAG=..." ...   ->  pointcloud_stitching_amd/lib/lab/libpcs_hip_<name>.so
set -e
cd "$(dirname "$0")/../../pointcloud_stitching_amd/csrc"
make -s >/dev/null
mkdir -p ../lib/lab
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
      -Wno-unused-parameter $defs -c pcs_voxel.hip -o /tmp/pcs_voxel_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/lab/libpcs_hip_$name.so pcs_kernels.o pcs_kernels_voxel.o /tmp/pcs_voxel_$name.o pcs_capi.o pcs_capi_voxel.o
  ls -la ../lib/lab/libpcs_hip_$name.so
done
