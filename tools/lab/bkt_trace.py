#!/usr/bin/env python3
"""lab: where a bucket-reduce workgroup's time goes. Needs a variant library built with -DPCS_BKT_TRACE (PCS_LIB_PATH);
thread 0 of every workgroup of pcs_vox_bkt_reduce_kernel stamps wall_clock64 (100 MHz) at its phase boundaries."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as Syn
from pointcloud_stitching_amd.api import PcsContext
S, W, H = 16, 1920, 1080
dev = torch.device("cuda", 0)
trace = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
os.environ["PCS_BKT_TRACE_PTR"] = str(trace.data_ptr())
cfgs = [Syn.synth_stream_config(W, H, s) for s in range(S)]
ctx = PcsContext(cfgs, flags=4)
dep = [torch.from_numpy(Syn.synth_depth(W, H, s).reshape(-1).view(np.uint8)).to(dev) for s in range(S)]
col = [torch.from_numpy(Syn.synth_color(W, H, s)).to(dev) for s in range(S)]
vox = torch.empty(S * W * H * 5, dtype=torch.int16, device=dev); nv = torch.zeros(2, dtype=torch.int32, device=dev)
for _ in range(6):
    ctx.process_frames_voxel_device([t.data_ptr() for t in dep], [t.data_ptr() for t in col], 50, vox.data_ptr(), vox.numel(), nv.data_ptr())
ctx.synchronize()
t = trace.cpu().numpy().reshape(1024, 16).astype(np.float64) / 100.0        # us
t0 = t[:, 0].min()
names = ["start->cleared (boff + first batch requested)", "->all partials in table", "->dense + published", "->sorted", "->base known", "->records written"]
print(f"voxels {int(nv[0])}; workgroup start times: first {0.0:.1f}, median {np.median(t[:,0]-t0):.1f}, last {(t[:,0]-t0).max():.1f} us; last end {(t[:,6]-t0).max():.1f} us")
for r, sel in (("first round (b < 512)", slice(0, 512)), ("second round (b >= 512)", slice(512, 1024))):
    d = np.diff(t[sel, :7], axis=1)
    print(r, "life median %.2f us" % np.median(t[sel, 6] - t[sel, 0]))
    for i, n in enumerate(names):
        print(f"   {n:50s} median {np.median(d[:, i]):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f} us")
    print("   inside 'sorted': wave sort %.2f, park + barrier %.2f, rank + srt + barrier %.2f us" % (
        np.median(t[sel, 7] - t[sel, 3]), np.median(t[sel, 8] - t[sel, 7]), np.median(t[sel, 4] - t[sel, 8])))
# who holds the others up: a bucket's first voxel is known once every lower-numbered bucket has published its count (stamp 3)
pub = t[:, 3] - t0
late = np.argsort(-pub[:512])[:6]
print("latest publishers of the first round:", [(int(b), round(float(t[b, 0] - t0), 1), round(float(pub[b]), 1)) for b in late], "(bucket, start, published at: us)")
st = t[:, 0] - t0
print("start time by bucket: b=0 %.1f, 63 %.1f, 127 %.1f, 255 %.1f, 383 %.1f, 511 %.1f, 512 %.1f, 767 %.1f, 1023 %.1f us" % tuple(st[[0, 63, 127, 255, 383, 511, 512, 767, 1023]]))
ctx.close()
