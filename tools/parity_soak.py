#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: random camera configurations (tests/test_gpu_parity.py::_random_config),
random and scene depth rasters, all flag/stride combinations, certified and forced-IEEE arithmetic — every
stitched buffer compared bit for bit with the oracle. Runs until the time budget is used up.

    python tools/parity_soak.py [seconds=240] [seed=1]
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
from pointcloud_stitching_amd.types import (FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID,  # noqa: E402
                                            FLAG_FORCE_IEEE, FLAG_TEXCOORD_HALF_PIXEL)
from tests.test_gpu_parity import _random_config                       # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
SIZES = [(64, 48), (128, 96), (104, 40), (200, 37), (320, 240), (640, 480), (424, 240)]
CSIZES = [(64, 48), (192, 108), (100, 75), (320, 180), (640, 480), (1280, 720)]
FLAGS = [0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_DROP_INVALID]
t0 = time.time()
trials = fails = 0
maths = {}
while time.time() - t0 < budget:
    wild = rng.random() < 0.4
    n_streams = int(rng.integers(1, 4))
    cfgs, depth, color = [], [], []
    for s in range(n_streams):
        w, h = SIZES[rng.integers(0, len(SIZES))]
        cw, ch = CSIZES[rng.integers(0, len(CSIZES))]
        cfgs.append(_random_config(rng, w, h, cw, ch, wild))
        sd = int(rng.integers(0, 1 << 30))
        depth.append(S.synth_depth(w, h, s, seed=sd, mode="random" if rng.random() < 0.4 else "scene"))
        color.append(S.synth_color(cw, ch, s, seed=sd))
    flags = FLAGS[rng.integers(0, len(FLAGS))] | (FLAG_TEXCOORD_HALF_PIXEL if rng.random() < 0.25 else 0)
    ds = int(rng.choice([1, 1, 1, 2, 3, 7]))
    want, wcounts = O.process_frames(cfgs, depth, color, flags, ds)
    for extra in (0, FLAG_FORCE_IEEE):
        with PcsContext(cfgs, flags=flags | extra, downsample=ds) as ctx:
            if not extra:
                for s in range(n_streams):
                    maths[ctx.stream_math(s)] = maths.get(ctx.stream_math(s), 0) + 1
            buf, counts, nbytes = ctx.process_frames(depth, color)
        got = buf[2:2 + nbytes // 2].reshape(-1, 5)
        if counts != wcounts or got.shape != want.shape or (got != want).any():
            fails += 1
            bad = np.argwhere(got != want)[:3] if got.shape == want.shape else "shape"
            print(f"MISMATCH trial {trials} seed {seed} wild {wild} flags {flags:#x} ieee {bool(extra)} ds {ds}: {bad}")
    trials += 1
print(f"soak: {trials} frame-sets, {fails} mismatches, stream_math histogram {dict(sorted(maths.items()))}, "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
