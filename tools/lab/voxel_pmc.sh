#!/bin/bash
# Runs on the GPU box: SQ counters of the voxel pre-aggregation kernels (payload reader and raster reader).  tools/voxel_pmc.sh [leaf]
LEAF=${1:-50}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT"; do
  rm -rf /tmp/vpmc
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/vpmc -- python $R/tools/voxel_bench.py 16 1920 1080 $LEAF > /tmp/vpmc.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/vpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("pcs::(anonymous namespace)::", "").replace("void ", "")
        if "partials" in n:
            agg[n[:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k, " ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(cs.items())))
PY
done
