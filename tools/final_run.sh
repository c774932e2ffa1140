#!/bin/bash
# Runs on the GPU box (via gpurun): the round's closing record on ONE build — GPU test-suite, randomised soaks, bench lines of every
# route, rocprofv3 stats + PMC (tools/profile_gpu.sh). Everything lands in gpurun_out/final_<tag>/ ; copy what is kept to profiles/.
#   tools/final_run.sh <tag> [soak_seconds=60]
TAG=${1:-r06}; SOAK=${2:-60}
OUT=$PWD/gpurun_out/final_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|^E " | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt
{
  timeout $((SOAK*3)) python tools/parity_soak.py $SOAK 7101
  timeout $((SOAK*3)) python tools/pack_soak.py $((SOAK/2)) 7102
  timeout $((SOAK*3)) python tools/batch_soak.py $((SOAK/2)) 7103
  timeout $((SOAK*3)) python tools/voxel_soak.py $SOAK 7104
  PCS_VOXEL_TAIL=bucket timeout $((SOAK*3)) python tools/voxel_soak.py $SOAK 7105
  timeout $((SOAK*3)) python tools/voxel_raster_soak.py $SOAK 7106
  PCS_VOXEL_TAIL=bucket timeout $((SOAK*3)) python tools/voxel_raster_soak.py $SOAK 7107
  timeout $((SOAK*4)) python tools/node_soak.py $SOAK 7108
} 2>&1 | grep -v -E "amdgpu.ids|RCCL version|HIP version|ROCm version|Hostname|Librccl" > $OUT/soak.log; cat $OUT/soak.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py > $OUT/bench_again.json 2>/dev/null
python bench.py --workload config5 --steps 60 --warmup 5 > $OUT/bench_config5.json 2>/dev/null
python bench.py --workload config5 --route node --gpus 1 --steps 200 --warmup 10 > $OUT/bench_config5_node1.json 2>/dev/null
python bench.py --workload config5 --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 5 > $OUT/bench_config5_node8v.json 2>/dev/null
python bench.py --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 10 > $OUT/bench_node8v.json 2>/dev/null
python bench.py --route node --gpus 1 --steps 200 --warmup 20 > $OUT/bench_node1.json 2>/dev/null
python bench.py --gpus 3 --steps 20 --warmup 5 > $OUT/bench_gpus3_folded.json 2>/dev/null
# the voxel pipeline by leaf: as shipped (warm bucket tail from 34 mm, LSD below), the bucket tail held to its cold chain, the LSD tail
for t in default cold lsd; do
  unset PCS_VOXEL_TAIL PCS_VOXEL_REGIONS
  if [ $t = lsd ]; then export PCS_VOXEL_TAIL=lsd; fi
  if [ $t = cold ]; then export PCS_VOXEL_REGIONS=0; fi
  python tools/voxel_probe.py 20,32,36,40,50,100,200 60 | grep leaf
done > $OUT/voxel_by_leaf.txt 2>&1
unset PCS_VOXEL_TAIL PCS_VOXEL_REGIONS
# BASELINE configs[4] as a frame loop: one context, two in turn (the tail of k beside the pre-aggregation of k+1), each with and without
# the colour-row table; BASELINE configs[1]: one 720p stream per launch, both tile shapes, and the a2 twin's one-cloud call
{ for c in 1 2; do python tools/voxel_overlap_probe.py $c 50 300; PCS_ROW_CONST=0 python tools/voxel_overlap_probe.py $c 50 300; done
  python tools/voxel_overlap_probe.py 2 200 300
  for t in 0 1; do PCS_SMALL_TILES=$t python tools/single_probe.py 1 5000; PCS_SMALL_TILES=$t python tools/single_probe.py 1 5000 twin; done
} 2>&1 | grep -E "context|per launch" > $OUT/probes.txt; cat $OUT/probes.txt
{ PCS_VOXEL_TAIL=bucket python tools/lab/bkt_cliff.py; PCS_VOXEL_TAIL=bucket PCS_VOXEL_REGIONS=0 python tools/lab/bkt_cliff.py | sed 's/\[bucket\]/[bucket, cold chain]/'; PCS_VOXEL_TAIL=lsd python tools/lab/bkt_cliff.py; } > $OUT/bkt_cliff.txt 2>&1
bash tools/profile_gpu.sh $TAG 200 > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
echo final_run done
