#!/usr/bin/env python3
"""lab: how full the buckets' regions get on warm calls (PCS_VOXEL_REGIONS=1). Needs a variant library built with -DPCS_BKT_TRACE
(PCS_LIB_PATH): it exports the workspace's address, whose first words are laid out independently of the cloud size —
ctl 2 x 64 words | published counts 1024 | cursors 2 x 1024 | {buckets, slots} 2 x 64 | splitters 1024 x 8 B."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("PCS_VOXEL_REGIONS", "1")
from pointcloud_stitching_amd import synthetic as Syn
from pointcloud_stitching_amd.api import PcsContext
leaves = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "40,50,100").split(",")]
S, W, H = 16, 1920, 1080
dev = torch.device("cuda", 0)
cfgs = [Syn.synth_stream_config(W, H, s) for s in range(S)]
ctx = PcsContext(cfgs, flags=4)
dep = [torch.from_numpy(Syn.synth_depth(W, H, s).reshape(-1).view(np.uint8)).to(dev) for s in range(S)]
col = [torch.from_numpy(Syn.synth_color(W, H, s)).to(dev) for s in range(S)]
vox = torch.empty(S * W * H * 5, dtype=torch.int16, device=dev); nv = torch.zeros(2, dtype=torch.int32, device=dev)
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
head = np.empty(13312 // 4, np.uint32)
for leaf in leaves:
    for it in range(5):
        ctx.process_frames_voxel_device([t.data_ptr() for t in dep], [t.data_ptr() for t in col], leaf, vox.data_ptr(), vox.numel(), nv.data_ptr())
        ctx.synchronize()
        base = int(libc.getenv(b"PCS_BKT_WS_BASE"))
        ctx.memcpy_d2h(head, base)
        ctl = head[:128].reshape(2, 64); cur = head[1152:3200].reshape(2, 1024); reg = head[3200:3328].reshape(2, 64)
        for par in (0, 1):
            c = cur[par]
            if c.any():
                B, cap = int(reg[par, 0]), int(reg[par, 1])
                cc = c[:max(B, 1)]
                print(f"leaf {leaf} call {it}: voxels {int(nv[0])}; cursors[{par}]: B {B} cap {cap} sum {int(c.sum())} mean {cc.mean():.0f} max {int(cc.max())} "
                      f"p99 {np.percentile(cc, 99):.0f} min {int(cc.min())} over-cap buckets {int((cc > cap).sum())} excess {int(np.maximum(cc.astype(np.int64) - cap, 0).sum())}; "
                      f"ctl0 {ctl[:, 0].tolist()} ctl3 {ctl[:, 3].tolist()}", flush=True)
        print(f"leaf {leaf} call {it}: next reg {reg[:, :2].tolist()}", flush=True)
ctx.close()
