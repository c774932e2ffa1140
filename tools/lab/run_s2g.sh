mkdir -p gpurun_out/s2g; O=gpurun_out/s2g
python -m pytest tests/test_node.py tests/test_voxel_stall.py tests/test_config5_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|^E " | tail -8 > $O/pytest.txt
for c in 0 1; do for i in a b; do
  PCS_NODE_COUNT_COPY=$c python bench.py --workload config5 --route node --gpus 1 --steps 300 --warmup 10 > $O/n1_copy${c}_$i.json 2>/dev/null
  PCS_NODE_COUNT_COPY=$c python bench.py --workload config5 --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 5 > $O/n8v_copy${c}_$i.json 2>/dev/null
done; done
