#!/usr/bin/env python3
"""Randomised soak of the voxel-grid pipeline on the GPU box: random payloads (uniform clouds, image-like surfaces with
long runs, clustered blobs, extreme coordinates), random sizes up to 400 k points, leaves from 1 mm to 32767 mm (half of the calls
with the leaf of the call before: warm calls of the bucket tail on an unrelated cloud), host and counted-device entry points,
aligned and 2-byte-skewed payloads — every result compared bit for bit with the oracle.

    python tools/voxel_soak.py [seconds=120] [seed=1]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pcs_oracle as O                                    # noqa: E402
from pointcloud_stitching_amd import synthetic as S                    # noqa: E402
from pointcloud_stitching_amd.api import PcsContext                    # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def payload(n):
    p = np.zeros((n, 5), np.int16)
    kind = rng.integers(0, 5)
    if kind == 0:                                   # uniform cube
        span = int(rng.choice([50, 600, 5000, 32767]))
        p[:, :3] = rng.integers(-span, span + 1, (n, 3))
    elif kind == 1:                                 # image-like rows: smooth x, constant-ish y, noisy z
        w = int(rng.choice([64, 640, 1920]))
        i = np.arange(n)
        p[:, 0] = ((i % w) * (6000 // w) - 3000).astype(np.int16)
        p[:, 1] = ((i // w) * 3 - 2000).astype(np.int16)
        p[:, 2] = 2000 + rng.integers(-8, 9, n)
    elif kind == 2:                                 # a few dense blobs: very long runs after the sort
        c = rng.integers(-30000, 30000, (int(rng.integers(1, 6)), 3))
        p[:, :3] = c[rng.integers(0, c.shape[0], n)] + rng.integers(-3, 4, (n, 3))
    elif kind == 3:                                 # extremes of the int16 range
        p[:, :3] = rng.choice(np.array([-32768, -32767, -1, 0, 1, 32766, 32767], np.int16), (n, 3))
    else:                                           # everything in one voxel
        p[:, :3] = rng.integers(100, 104, (n, 3))
    p[:, 3] = rng.integers(0, 65536, n, dtype=np.uint16).view(np.int16)
    p[:, 4] = rng.integers(0, 256, n)
    return p


cfgs, _, _ = S.synth_frame_set(1, 64, 48)
t0 = time.time()
trials = fails = 0
with PcsContext(cfgs) as ctx:
    cap = 400_000
    d_in = ctx.device_malloc(cap * 10 + 64)
    d_out = ctx.device_malloc(cap * 10 + 64)
    d_n = ctx.device_malloc(4)
    d_nv = ctx.device_malloc(4)
    while time.time() - t0 < budget:
        n = int(rng.integers(1, cap)) if rng.random() < 0.5 else int(rng.choice([1, 2, 3, 255, 256, 257, 8191, 8192, 8193]))
        leaf = int(rng.choice([1, 2, 7, 10, 50, 64, 200, 1000, 5000, 32767, int(rng.integers(1, 32768))]))
        if trials and rng.random() < 0.5:
            leaf = prev_leaf            # the same leaf as the call before: a WARM call of the bucket tail (regions sized and split by
        prev_leaf = leaf                # the previous, unrelated cloud: overflow into the general list, gathers, split buckets)
        p = payload(n)
        want = O.voxel_grid(p, leaf)
        mode = rng.integers(0, 3)
        if mode == 0:
            got = ctx.voxel_grid(p, leaf)
        else:
            skew = 2 if mode == 2 else 0
            ctx.memcpy_h2d(d_in + skew, p)
            ctx.memcpy_h2d(d_n, np.array([n], np.int32))
            live_cap = min(cap, n + int(rng.integers(0, 5000)))
            ctx.voxel_grid_device_counted(d_in + skew, d_n, live_cap, leaf, d_out, cap * 5, d_nv)
            ctx.synchronize()
            nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
            got = np.empty((int(nv[0]), 5), np.int16)
            if nv[0]:
                ctx.memcpy_d2h(got, d_out)
        if got.shape != want.shape or (got != want).any():
            fails += 1
            print(f"MISMATCH trial {trials} seed {seed}: n {n} leaf {leaf} mode {mode}: got {got.shape} want {want.shape}")
        trials += 1
print(f"voxel soak: {trials} clouds, {fails} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
