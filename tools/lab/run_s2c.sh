mkdir -p gpurun_out/s2c; O=gpurun_out/s2c
python -m pytest tests/test_config5_sharded.py tests/test_node.py tests/test_voxel_stall.py -m gpu -x -q 2>&1 | grep -v -E "amdgpu.ids|RCCL version|HIP version|ROCm version|Hostname|Librccl" | tail -15 > $O/pytest.txt
N8="python bench.py --workload config5 --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 5"
$N8 > $O/n8v.json 2> $O/n8v.err
$N8 > $O/n8v_b.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 $N8 > $O/n8v_q8.json 2>/dev/null
python bench.py --workload config5 --route node --gpus 1 --steps 200 --warmup 10 > $O/n1.json 2>/dev/null
python bench.py --workload config5 --gpus 4 --node-devices 0,0,0,0 --steps 100 --warmup 5 > $O/n4v.json 2>/dev/null
python bench.py --workload config5 --gpus 2 --node-devices 0,0 --steps 100 --warmup 5 > $O/n2v.json 2>/dev/null
