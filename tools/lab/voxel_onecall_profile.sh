R=$PWD
cd /tmp && export TMPDIR=/tmp
cat > /tmp/onecall.py <<PY
import sys, os
sys.path.insert(0, "$R")
import numpy as np
from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
from pointcloud_stitching_amd.types import FLAG_DROP_INVALID
n, W, H, leaf = 16, 1920, 1080, int(sys.argv[1])
cfgs = [S.synth_stream_config(W, H, s) for s in range(n)]
with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
    dd, dc = [], []
    for s in range(n):
        d = S.synth_depth(W, H, s); c = S.synth_color(W, H, s)
        p = ctx.device_malloc(d.nbytes + 256); ctx.memcpy_h2d(p, d); dd.append(p)
        p = ctx.device_malloc(c.nbytes + 256); ctx.memcpy_h2d(p, c); dc.append(p)
    n_max = n * W * H
    d_vox = ctx.device_malloc(n_max * 10 + 64); d_nv = ctx.device_malloc(4)
    for _ in range(30):
        ctx.process_frames_voxel_device(dd, dc, leaf, d_vox, n_max * 5, d_nv)
    ctx.synchronize()
PY
for leaf in $*; do
rm -rf /tmp/vp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -- python /tmp/onecall.py $leaf > /tmp/vp.log 2>&1
echo leaf $leaf
python - <<PY
import csv, glob
tot=0
for f in glob.glob("/tmp/vp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("pcs::(anonymous namespace)::", "")
        if "voxel" in n:
            per=float(r["TotalDurationNs"])/30/1e3
            tot+=per
            print(n[:52].ljust(52), r["Calls"].rjust(6), "%9.2f us avg" % (float(r["AverageNs"]) / 1e3), "%9.2f us per call" % per)
print("sum of kernel time per call %.1f us" % tot)
PY
done
