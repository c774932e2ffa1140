#!/usr/bin/env python3
"""Times the config-5 tail on one GPU: fused kernel with invalid-depth compaction on 16 x 1920x1080 streams,
then the voxel-grid downsample of the stitched cloud (device-resident end to end)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_stitching_amd import synthetic as S          # noqa: E402
from pointcloud_stitching_amd.api import PcsContext          # noqa: E402
from pointcloud_stitching_amd.types import FLAG_DROP_INVALID  # noqa: E402

n_streams, W, H = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 1920, 1080)
leafs = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [10, 50, 200]
cfgs = [S.synth_stream_config(W, H, s) for s in range(n_streams)]
with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
    n_max = n_streams * W * H
    slab = ctx.device_malloc(n_streams * (W * H * 5 + 1024) + 2 * (n_max * 10 + 256) + 4096)
    off = 0
    dd, dc = [], []
    for s in range(n_streams):
        d = S.synth_depth(W, H, s); c = S.synth_color(W, H, s)
        dd.append(slab + off); ctx.memcpy_h2d(slab + off, d); off += (d.nbytes + 255) & ~255
        dc.append(slab + off); ctx.memcpy_h2d(slab + off, c); off += (c.nbytes + 16 + 255) & ~255
    d_pay = slab + off; off += (n_max * 10 + 255) & ~255
    d_vox = slab + off
    d_cnt = ctx.device_malloc(4 * (n_streams + 1))
    d_nv = ctx.device_malloc(4)
    for _ in range(3):
        ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
    ctx.synchronize()
    cnt = np.empty(n_streams + 1, np.int32); ctx.memcpy_d2h(cnt, d_cnt)
    total = int(cnt[-1])
    reps = 20
    ctx.timer_begin()
    for _ in range(reps):
        ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
    ctx.timer_end()
    t_pack = ctx.timer_elapsed_ms() / reps
    print(f"{n_streams} x {W}x{H}: {n_max} pixels -> {total} points kept; compaction pack {t_pack:.3f} ms "
          f"({n_max / t_pack / 1e3:.0f} Mpoints/s)")
    for leaf in leafs:
        for _ in range(2):
            ctx.voxel_grid_device(d_pay, total, leaf, d_vox, n_max * 5, d_nv)
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(reps):
            ctx.voxel_grid_device(d_pay, total, leaf, d_vox, n_max * 5, d_nv)
        ctx.timer_end()
        t = ctx.timer_elapsed_ms() / reps
        nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
        # the whole of config 5, device-resident and asynchronous: compaction launch(es), then the counted voxel grid
        ctx.timer_begin()
        for _ in range(reps):
            ctx.process_frames_device(dd, dc, d_pay, n_max * 5, d_cnt)
            ctx.voxel_grid_device_counted(d_pay, d_cnt + 4 * n_streams, n_max, leaf, d_vox, n_max * 5, d_nv)
        ctx.timer_end()
        t_all = ctx.timer_elapsed_ms() / reps
        for _ in range(2):
            ctx.process_frames_voxel_device(dd, dc, leaf, d_vox, n_max * 5, d_nv)
        ctx.synchronize()
        nv2 = np.empty(1, np.int32); ctx.memcpy_d2h(nv2, d_nv)
        assert int(nv2[0]) == int(nv[0]), (nv2, nv)
        ctx.timer_begin()
        for _ in range(reps):
            ctx.process_frames_voxel_device(dd, dc, leaf, d_vox, n_max * 5, d_nv)
        ctx.timer_end()
        t_one = ctx.timer_elapsed_ms() / reps
        print(f"  rasters -> voxels in one call (no stitched cloud): {t_one:.3f} ms per frame-set ({n_max / t_one / 1e3:.0f} Mpixels/s)")
        print(f"  voxel grid leaf {leaf:4d} mm: {t:.3f} ms ({total / t / 1e3:.0f} Mpoints/s in) -> {int(nv[0])} voxels; "
              f"compaction + stitch + voxel grid, no host sync: {t_all:.3f} ms per frame-set ({n_max / t_all / 1e3:.0f} Mpixels/s)")
