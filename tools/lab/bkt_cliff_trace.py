#!/usr/bin/env python3
"""lab: bkt_cliff.py's stale call (cloud A on splitters made for B) with the reduce kernel's phase stamps (variant library built
with -DPCS_BKT_TRACE, PCS_LIB_PATH): which workgroups are slow, and where."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
dev = torch.device("cuda", 0)
trace = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
os.environ["PCS_BKT_TRACE_PTR"] = str(trace.data_ptr())
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
    for name in ("B", "B", "B", "A", "A"):
        d, n = bufs[name]
        trace.zero_(); torch.cuda.synchronize()
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.voxel_grid_device(d, 400000 if name == "B" else n, 40, out, 400000 * 5, cnt) if False else ctx.voxel_grid_device(d, n, 40, out, 400000 * 5, cnt)
        ctx.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        t = trace.cpu().numpy().reshape(1024, 16).astype(np.float64) / 100.0
        live = t[:, 0] > 0
        if live.any():
            z = t[live, 0].min()
            life = np.where(live, t[:, 6] - t[:, 0], 0)
            top = np.argsort(-life)[:4]
            print(f"{name}: {ms:8.3f} ms; slowest workgroups (bucket: start, stamps 1..8 relative to start, us):")
            for w in top:
                print("   ", int(w), round(float(t[w, 0] - z), 1), [round(float(t[w, i] - t[w, 0]), 1) for i in range(1, 9)])
        else:
            print(f"{name}: {ms:8.3f} ms")
