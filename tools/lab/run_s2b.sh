mkdir -p gpurun_out/s2b; O=gpurun_out/s2b
N8="python bench.py --workload config5 --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 5"
for q in default 8 16 24; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  $N8 > $O/n8v_q$q.json 2>/dev/null
  $N8 > $O/n8v_q${q}_b.json 2>/dev/null
done
unset GPU_MAX_HW_QUEUES
python tools/single_probe.py 1 5000 > $O/single_probe.txt 2>&1
python bench.py --no-config5 --no-host-api --no-cpu-baseline > $O/bench_lite.json 2> $O/bench_lite.err
