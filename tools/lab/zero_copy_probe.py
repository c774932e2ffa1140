#!/usr/bin/env python3
"""Probe: the fused kernel reading its rasters from / writing its payload to PINNED HOST memory directly (zero copy), vs
the staged host API. 8 x 1280x720.   tools/zero_copy_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloud_stitching_amd import synthetic as S          # noqa: E402
from pointcloud_stitching_amd.api import PcsContext          # noqa: E402
from pointcloud_stitching_amd.types import POINT_SHORTS      # noqa: E402
from oracle import pcs_oracle as O                           # noqa: E402

n, W, H = 8, 1280, 720
cfgs, depth, color = S.synth_frame_set(n, W, H)
n_sh = n * W * H * POINT_SHORTS
with PcsContext(cfgs) as ctx:
    keep = []

    def pinned(nbytes):
        a = ctx.host_array((nbytes,), np.uint8)
        keep.append(a)
        return a.ctypes.data
    hd = [pinned(d.nbytes) for d in depth]
    hc = [pinned(c.nbytes) for c in color]
    for p, a in zip(hd + hc, depth + color):
        C.memmove(p, a.ctypes.data, a.nbytes)
    hout = pinned(n_sh * 2 + 64)
    dd = [ctx.device_malloc(d.nbytes) for d in depth]
    dc = [ctx.device_malloc(c.nbytes) for c in color]
    for p, a in zip(dd + dc, depth + color):
        ctx.memcpy_h2d(p, a)
    dout = ctx.device_malloc(n_sh * 2 + 64)
    want, _ = O.process_frames(cfgs, depth, color)

    def run(dp, cp, outp, reps=10):
        ctx.process_frames_device(dp, cp, outp, n_sh); ctx.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            ctx.process_frames_device(dp, cp, outp, n_sh)
        ctx.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    for name, dp, cp, outp in (("all in HBM", dd, dc, dout), ("rasters from pinned host", hd, hc, dout),
                               ("payload to pinned host", dd, dc, hout), ("both over PCIe (zero copy)", hd, hc, hout)):
        ms = run(dp, cp, outp)
        print(f"{name:32s} {ms:8.3f} ms per frame-set")
    got = np.frombuffer((C.c_int16 * n_sh).from_address(hout), dtype=np.int16).reshape(-1, 5)
    print("zero-copy payload equals the oracle:", bool((got == want).all()))
