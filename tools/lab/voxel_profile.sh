#!/bin/bash
# Runs on the GPU box: per-kernel times of the config-5 tail (tools/voxel_bench.py) under rocprofv3.  tools/voxel_profile.sh [leafs]
LEAFS=${1:-50}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -- python $R/tools/voxel_bench.py 16 1920 1080 $LEAFS > /tmp/vp.log 2>&1
grep "voxel grid\|compaction" /tmp/vp.log
python - <<PY
import csv, glob
for f in glob.glob("/tmp/vp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("pcs::(anonymous namespace)::", "")
        if "voxel" in n:
            print(n[:52].ljust(52), r["Calls"].rjust(6), "%9.2f us" % (float(r["AverageNs"]) / 1e3))
PY
