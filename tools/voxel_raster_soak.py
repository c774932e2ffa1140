#!/usr/bin/env python3
"""Randomised soak of pcs_process_frames_voxel_device (rasters -> voxel grid, no stitched cloud) against the oracle's
voxel grid over the oracle's stitched cloud.   tools/voxel_raster_soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_stitching_amd import synthetic as S                                      # noqa: E402
from pointcloud_stitching_amd.api import PcsContext                                      # noqa: E402
from pointcloud_stitching_amd.types import (FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID, FLAG_FORCE_IEEE,  # noqa: E402
                                            FLAG_TEXCOORD_HALF_PIXEL)
from oracle import pcs_oracle as O                                                       # noqa: E402
from tests.test_gpu_parity import _random_config                                         # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
runs = patch_runs = bad = 0
while time.time() - t0 < budget:
    n = int(rng.integers(1, 5))
    shapes = []
    for s in range(n):
        w = int(rng.choice([8, 16, 64, 72, 128, 200, 320, 333, 640, 848, 1280]) if rng.random() < 0.7 else rng.integers(2, 900))
        h = int(rng.choice([1, 2, 63, 64, 65, 129, 240, 480]) if rng.random() < 0.7 else rng.integers(1, 500))
        shapes.append((w, h))
    patch = rng.random() < 0.6
    if patch:
        shapes = [(max(8, w - w % 8), h) for w, h in shapes]
    flags = int(rng.choice([0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_DROP_INVALID,
                            FLAG_DROP_INVALID | FLAG_FORCE_IEEE]))
    stride = int(rng.choice([1, 1, 1, 1, 2, 5]))
    # a third of the frame-sets use random cameras (distortion models, depth->colour rotations, colour rasters of another
    # size, configurations the arithmetic certificate must refuse) instead of the synthetic rig
    rand_cam = rng.random() < 0.33
    csizes = [(int(rng.choice([64, 100, 192, 320])), int(rng.choice([48, 75, 108, 180]))) if rand_cam else (w, h) for (w, h) in shapes]
    if rand_cam:
        cfgs = [_random_config(rng, w, h, cw, ch, rng.random() < 0.4) for (w, h), (cw, ch) in zip(shapes, csizes)]
        if rng.random() < 0.3:
            flags |= FLAG_TEXCOORD_HALF_PIXEL
    else:
        cfgs = [S.synth_stream_config(w, h, int(rng.integers(0, 8))) for (w, h) in shapes]
    depth, color = [], []
    for s, (w, h) in enumerate(shapes):
        kind = rng.random()
        if kind < 0.6:
            d = S.synth_depth(w, h, s, seed=int(rng.integers(1, 1 << 30)))
        elif kind < 0.8:
            d = rng.integers(0, 65536, w * h, dtype=np.uint16)
        else:
            d = np.full(w * h, int(rng.integers(0, 3000)), np.uint16)
        depth.append(np.ascontiguousarray(d, np.uint16).reshape(-1))
        color.append(S.synth_color(csizes[s][0], csizes[s][1], s, seed=int(rng.integers(1, 1 << 30))))
    n_max = sum(c.n_points for c in cfgs)
    stitched, _ = O.process_frames(cfgs, depth, color, flags, stride)
    with PcsContext(cfgs, flags=flags, downsample=stride) as ctx:
        dd = [ctx.device_malloc(max(d.nbytes, 16)) for d in depth]
        dc = [ctx.device_malloc(max(c.nbytes, 16)) for c in color]
        for ptr, a in zip(dd + dc, depth + color):
            ctx.memcpy_h2d(ptr, a)
        d_vox = ctx.device_malloc(n_max * 10 + 64)
        d_nv = ctx.device_malloc(4)
        for leaf in rng.choice([1, 2, 5, 13, 29, 30, 36, 50, 77, 200, 1000, 32767], 3, replace=False):
            leaf = int(leaf)
            ctx.process_frames_voxel_device(dd, dc, leaf, d_vox, n_max * 5, d_nv)
            ctx.synchronize()
            nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
            got = np.empty(max(int(nv[0]), 1) * 5, np.int16); ctx.memcpy_d2h(got, d_vox)
            got = got[:int(nv[0]) * 5].reshape(-1, 5)
            want = O.voxel_grid(stitched, leaf)
            runs += 1
            patch_runs += int(patch and stride == 1)
            if got.shape != want.shape or (got != want).any():
                bad += 1
                print("MISMATCH", shapes, flags, stride, leaf, got.shape, want.shape, flush=True)
print(f"{runs} raster->voxel calls ({patch_runs} through the square-patch reader), {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
