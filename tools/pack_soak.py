#!/usr/bin/env python3
"""Adversarial soak of the a2 twin (pcs_copy_pointcloud_xyzrgb_to_buffer) on the GPU box: vertices and texcoords
drawn from ALL float32 bit patterns (NaN, infinities, denormals, huge), mixed with realistic ones, random
extrinsics incl. extreme entries, every flag combination — bit for bit against the oracle.

    python tools/pack_soak.py [seconds=120] [seed=1]
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
from pointcloud_stitching_amd.types import FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
FLAGS = [0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_DROP_INVALID]
t0 = time.time()
trials = fails = 0
while time.time() - t0 < budget:
    cw, ch = [(64, 48), (640, 480), (1280, 720), (100, 75)][rng.integers(0, 4)]
    sc = S.synth_stream_config(64, 48, int(rng.integers(0, 8)), color_size=(cw, ch))
    kind = rng.integers(0, 3)
    if kind == 1:                       # extreme extrinsic entries
        for k in range(12):
            sc.cam_to_world[k] = float(np.float32(rng.normal() * 10.0 ** rng.integers(-20, 20)))
    elif kind == 2:                     # arbitrary bit patterns in the extrinsic
        bits = rng.integers(0, 1 << 32, 12, dtype=np.uint64).astype(np.uint32)
        for k in range(12):
            sc.cam_to_world[k] = float(bits[k:k + 1].view(np.float32)[0])
    n = int(rng.choice([1, 7, 8, 63, 2048, 2049, 10000, 70001]))
    vtx = rng.integers(0, 1 << 32, (n, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)
    tex = rng.integers(0, 1 << 32, (n, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
    real = rng.random(n) < 0.5          # half of the points look like a camera's
    vtx[real] = rng.normal(0, 2, (int(real.sum()), 3)).astype(np.float32)
    tex[real] = rng.uniform(-0.2, 1.2, (int(real.sum()), 2)).astype(np.float32)
    color = S.synth_color(cw, ch, 0, seed=int(rng.integers(0, 1 << 30)))
    flags = FLAGS[rng.integers(0, len(FLAGS))]
    want = O.pack(sc, vtx, tex, color, flags)
    with PcsContext([sc], flags=flags) as ctx:
        got, cnt = ctx.copy_pointcloud_xyzrgb_to_buffer(0, vtx, tex, color)
    if cnt != want.shape[0] or (got != want).any():
        fails += 1
        print(f"MISMATCH trial {trials} seed {seed} kind {kind} n {n} flags {flags:#x}")
    trials += 1
print(f"pack soak: {trials} calls, {fails} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
