#!/bin/bash
# GPU box: voxel_probe under each lab variant of libpcs_hip (tools/lab/build_vox_variants.sh), then under the shipped library.
#   tools/lab/vox_variants_run.sh "leaves" reps name1 name2 ...
LEAVES=$1; REPS=$2; shift 2
for v in "$@" shipped; do
  unset PCS_LIB_PATH
  if [ $v != shipped ]; then export PCS_LIB_PATH=$PWD/pointcloud_stitching_amd/lib/lab/libpcs_hip_$v.so; fi
  echo "== $v"; python tools/voxel_probe.py $LEAVES $REPS 2>&1 | grep leaf
done
