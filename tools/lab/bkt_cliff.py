#!/usr/bin/env python3
"""lab: what a call costs when the bucket tail's splitters are stale: cloud A (two far clusters), then cloud B (400 k voxels
inside ONE of A's key ranges), then B again (splitters refreshed), under PCS_VOXEL_TAIL=bucket / lsd."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
rng = np.random.default_rng(91)
def cloud(n, lo, hi):
    p = np.zeros((n, 5), np.int16); p[:, :3] = rng.integers(lo, hi, (n, 3)); return p
a = np.concatenate([cloud(60000, -30000, -29000), cloud(60000, 29000, 30000)])
b = cloud(400000, -4000, 4000)
cfgs, _, _ = S.synth_frame_set(1, 64, 48)
with PcsContext(cfgs) as ctx:
    bufs = {}
    for name, p in (("A", a), ("B", b)):
        d = ctx.device_malloc(p.nbytes + 64); ctx.memcpy_h2d(d, p); bufs[name] = (d, p.shape[0])
    out = ctx.device_malloc(b.nbytes + 64); cnt = ctx.device_malloc(4)
    for name in ("A", "A", "B", "B", "B", "A", "A"):
        d, n = bufs[name]
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.voxel_grid_device(d, n, 40, out, 400000 * 5, cnt)
        ctx.synchronize()
        print(f"[{os.environ.get('PCS_VOXEL_TAIL', 'default')}] {name}: {1e3 * (time.perf_counter() - t0):8.3f} ms", flush=True)
