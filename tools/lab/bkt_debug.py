#!/usr/bin/env python3
"""lab: one 1920x1080 stream, leaf 1 mm (every point its own voxel: every bucket of the bucket tail overflows) — time + count."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
leaf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfgs = [S.synth_stream_config(1920, 1080, 0)]
depth = [S.synth_depth(1920, 1080, 0)]; color = [S.synth_color(1920, 1080, 0)]
n = cfgs[0].n_points
with PcsContext(cfgs) as ctx:
    dd = ctx.device_malloc(depth[0].nbytes); dc = ctx.device_malloc(color[0].nbytes)
    ctx.memcpy_h2d(dd, depth[0]); ctx.memcpy_h2d(dc, color[0])
    d_vox = ctx.device_malloc(n * 10 + 64); d_nv = ctx.device_malloc(4)
    for it in range(3):
        t0 = time.perf_counter()
        ctx.process_frames_voxel_device([dd], [dc], leaf, d_vox, n * 5, d_nv)
        ctx.synchronize()
        nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
        print(f"leaf {leaf}: call {it}: {1e3 * (time.perf_counter() - t0):.2f} ms, voxels {int(nv[0])}", flush=True)
