mkdir -p gpurun_out/s2d; O=gpurun_out/s2d
N8="python bench.py --workload config5 --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 5"
for k in 1 2 3 4 8; do
  PCS_NODE_SINK_STREAMS=$k $N8 > $O/n8v_k$k.json 2>/dev/null
  PCS_NODE_SINK_STREAMS=$k $N8 > $O/n8v_k${k}_b.json 2>/dev/null
done
python -m pytest tests/test_config5_sharded.py tests/test_node.py tests/test_voxel_stall.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | grep -v -E "amdgpu.ids|RCCL version|HIP version|ROCm version|Hostname|Librccl" | tail -15 > $O/pytest.txt
