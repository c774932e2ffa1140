mkdir -p gpurun_out/s2i; O=gpurun_out/s2i
{ for rep in 1 2; do
  bash tools/lab/vox_variants_run.sh 40,50,100,200 120 nopre
  echo "== PCS_ROW_CONST=0"; PCS_ROW_CONST=0 bash tools/lab/vox_variants_run.sh 50,100 120 nopre
done; } > $O/ab.txt 2>&1
python -m pytest tests/test_voxel_grid.py tests/test_config5_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E " | tail -5 > $O/pytest.txt
{ python tools/voxel_overlap_probe.py 2 50 300; PCS_LIB_PATH=$PWD/pointcloud_stitching_amd/lib/lab/libpcs_hip_nopre.so python tools/voxel_overlap_probe.py 2 50 300; python tools/voxel_overlap_probe.py 2 50 300; PCS_LIB_PATH=$PWD/pointcloud_stitching_amd/lib/lab/libpcs_hip_nopre.so python tools/voxel_overlap_probe.py 2 50 300; } 2>&1 | grep context > $O/loop.txt
