#!/bin/bash
# GPU box: SQ counters of the raster reader of the voxel pipeline on one warm call of 16 x 1080p (tools/voxel_probe.py), three passes.
#   tools/lab/vox_frontend_pmc.sh [leaf=50]      (PCS_LIB_PATH selects a lab variant)
LEAF=${1:-50}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  rm -rf /tmp/vpmc
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/vpmc -- python $R/tools/voxel_probe.py $LEAF 20 > /tmp/vpmc.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/vpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("pcs::(anonymous namespace)::", "").replace("void ", "")
        if "partials" in n or "reduce" in n:
            agg[n[:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
done
