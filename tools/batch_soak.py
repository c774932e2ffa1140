#!/usr/bin/env python3
"""Randomised soak of pcs_process_frames_device_batch (K frame-sets per call; dense and ordered-compaction forms) against
the oracle.   tools/batch_soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_stitching_amd import synthetic as S                                      # noqa: E402
from pointcloud_stitching_amd.api import PcsContext                                      # noqa: E402
from pointcloud_stitching_amd.types import (FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID, FLAG_FORCE_IEEE,  # noqa: E402
                                            POINT_SHORTS)
from oracle import pcs_oracle as O                                                       # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
calls = sets_checked = bad = 0
while time.time() - t0 < budget:
    n = int(rng.choice([1, 2, 3, 5, 8, 16, 17]))
    shapes = [(int(rng.choice([8, 64, 72, 160, 320, 333, 640])), int(rng.choice([1, 2, 47, 48, 120, 240]))) for _ in range(n)]
    shapes = [(max(w, 2) if h == 1 else w, h) for w, h in shapes]
    flags = int(rng.choice([0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_DROP_INVALID,
                            FLAG_DROP_INVALID | FLAG_FORCE_IEEE]))
    stride = int(rng.choice([1, 1, 1, 3]))
    k_sets = int(rng.integers(1, 20))
    cfgs = [S.synth_stream_config(w, h, int(rng.integers(0, 8))) for (w, h) in shapes]
    sets = []
    for k in range(k_sets):
        dep, col = [], []
        for s, (w, h) in enumerate(shapes):
            kind = rng.random()
            d = (S.synth_depth(w, h, s, seed=int(rng.integers(1, 1 << 30))) if kind < 0.7 else
                 rng.integers(0, 65536, w * h, dtype=np.uint16) if kind < 0.85 else np.zeros(w * h, np.uint16))
            dep.append(np.ascontiguousarray(d, np.uint16).reshape(-1))
            col.append(S.synth_color(w, h, s, seed=int(rng.integers(1, 1 << 30))))
        sets.append((dep, col))
    with PcsContext(cfgs, flags=flags, downsample=stride) as ctx:
        n_sh = sum(c.n_points for c in cfgs) * POINT_SHORTS
        dd, dc = [], []
        for dep, col in sets:
            pd = [ctx.device_malloc(max(a.nbytes, 16)) for a in dep]
            pc = [ctx.device_malloc(max(a.nbytes, 16)) for a in col]
            for ptr, a in zip(pd + pc, dep + col):
                ctx.memcpy_h2d(ptr, a)
            dd.append(pd); dc.append(pc)
        skew = int(rng.choice([0, 0, 2, 4]))
        outs = [ctx.device_malloc(n_sh * 2 + 64) + skew for _ in range(k_sets)]
        d_counts = [ctx.device_malloc(4 * (n + 1)) if rng.random() < 0.7 else None for _ in range(k_sets)]
        ctx.process_frames_device_batch(dd, dc, outs, n_sh, d_counts)
        ctx.synchronize()
        calls += 1
        for k in rng.choice(k_sets, min(k_sets, 4), replace=False):
            k = int(k)
            want, wcounts = O.process_frames(cfgs, sets[k][0], sets[k][1], flags, stride)
            ok = True
            if d_counts[k] is not None:
                cnt = np.empty(n + 1, np.int32); ctx.memcpy_d2h(cnt, d_counts[k])
                ok = list(cnt[:n]) == list(wcounts) and int(cnt[n]) == sum(wcounts)
            got = np.empty(max(want.size, 1), np.int16); ctx.memcpy_d2h(got, outs[k])
            ok = ok and (got[:want.size].reshape(-1, 5) == want).all()
            sets_checked += 1
            if not ok:
                bad += 1
                print("MISMATCH", shapes, flags, stride, k_sets, k, flush=True)
print(f"{calls} batch calls, {sets_checked} frame-sets checked, {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
