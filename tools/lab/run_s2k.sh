mkdir -p gpurun_out/s2k; O=gpurun_out/s2k
{ for rep in 1 2; do for v in nopre shipped; do
  unset PCS_LIB_PATH; if [ $v != shipped ]; then export PCS_LIB_PATH=$PWD/pointcloud_stitching_amd/lib/lab/libpcs_hip_$v.so; fi
  echo "== $v"; python tools/voxel_bench.py 16 1920 1080 20,50,200 2>&1 | grep -v amdgpu.ids
done; done; } > $O/ab.txt 2>&1
unset PCS_LIB_PATH
python -m pytest tests/test_voxel_grid.py tests/test_config5_sharded.py tests/test_voxel_stall.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E " | tail -5 > $O/pytest.txt
