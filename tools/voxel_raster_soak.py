#!/usr/bin/env python3
"""Randomised soak of pcs_process_frames_voxel_device (rasters -> voxel grid, no stitched cloud) and of the partials exchange
format (cameras split over several contexts, partials merged on one, or pre-aggregated into one context's voxel sink) against
the oracle's voxel grid over the oracle's stitched cloud.   tools/voxel_raster_soak.py [seconds] [seed]"""
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
runs = patch_runs = shard_runs = sink_runs = bad = 0
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
            # the leaf's first call is the bucket tail's cold chain; then (two times in three) one camera sees something else —
            # another scene, noise, a wall — and the same leaf is called again: a WARM call on a cloud that moved, and once more
            # on the same rasters (steady state)
            scenes = [(stitched, None)]
            if rng.random() < 0.67:
                s2 = int(rng.integers(0, n))
                w2, h2 = shapes[s2]
                kind = rng.random()
                d2 = (S.synth_depth(w2, h2, s2, seed=int(rng.integers(1, 1 << 30))) if kind < 0.5 else
                      rng.integers(0, 65536, w2 * h2, dtype=np.uint16) if kind < 0.75 else np.full(w2 * h2, int(rng.integers(0, 6000)), np.uint16))
                depth2 = list(depth); depth2[s2] = np.ascontiguousarray(d2, np.uint16).reshape(-1)
                st2, _ = O.process_frames(cfgs, depth2, color, flags, stride)
                scenes += [(st2, (s2, depth2[s2])), (st2, None)]
            for want_cloud, change in scenes:
                if change is not None:
                    ctx.memcpy_h2d(dd[change[0]], change[1])
                ctx.process_frames_voxel_device(dd, dc, leaf, d_vox, n_max * 5, d_nv)
                ctx.synchronize()
                nv = np.empty(1, np.int32); ctx.memcpy_d2h(nv, d_nv)
                got = np.empty(max(int(nv[0]), 1) * 5, np.int16); ctx.memcpy_d2h(got, d_vox)
                got = got[:int(nv[0]) * 5].reshape(-1, 5)
                want = O.voxel_grid(want_cloud, leaf)
                runs += 1
                patch_runs += int(patch and stride == 1)
                if got.shape != want.shape or (got != want).any():
                    bad += 1
                    print("MISMATCH", shapes, flags, stride, leaf, got.shape, want.shape, flush=True)
            if len(scenes) > 1:      # back to the first scene for the next leaf (and for the sharded merge below)
                ctx.memcpy_h2d(dd[s2], depth[s2])
    # the exchange format (config 5 sharded): the same cameras split over 1..n contexts, every shard's partials appended to the
    # root's arrays in a random shard order, one sort + segmented mean over all of them
    if n >= 1:
        k = int(rng.integers(1, n + 1))
        cuts = sorted(rng.choice(np.arange(1, n), k - 1, replace=False).tolist()) if k > 1 else []
        bounds = [0] + cuts + [n]
        leaf = int(rng.choice([1, 7, 29, 36, 50, 200, 4000]))
        want = O.voxel_grid(stitched, leaf)
        ctxs = [PcsContext(cfgs[a:b], flags=flags, downsample=stride) for a, b in zip(bounds[:-1], bounds[1:])]
        try:
            root = ctxs[0]
            caps = [c.max_payload_shorts // 5 for c in ctxs]
            d_k = root.device_malloc(sum(caps) * 8 + 64); d_p = root.device_malloc(sum(caps) * 32 + 64)
            off = 0
            for r in rng.permutation(len(ctxs)):
                c = ctxs[r]; a, b = bounds[r], bounds[r + 1]
                dd = [c.device_malloc(max(d.nbytes, 16)) for d in depth[a:b]]
                dc = [c.device_malloc(max(x.nbytes, 16)) for x in color[a:b]]
                for ptr, arr in zip(dd + dc, depth[a:b] + color[a:b]):
                    c.memcpy_h2d(ptr, arr)
                k_r = c.device_malloc(caps[r] * 8 + 64); p_r = c.device_malloc(caps[r] * 32 + 64); n_r = c.device_malloc(64)
                c.process_frames_voxel_partials_device(dd, dc, leaf, k_r, p_r, caps[r], n_r)
                c.synchronize()
                m = np.empty(1, np.int32); c.memcpy_d2h(m, n_r); m = int(m[0])
                if m:
                    hk = np.empty(m, np.uint64); hp = np.empty(m * 8, np.uint32)
                    c.memcpy_d2h(hk, k_r); c.memcpy_d2h(hp, p_r)
                    root.memcpy_h2d(d_k + off * 8, hk); root.memcpy_h2d(d_p + off * 32, hp)
                off += m
            d_o = root.device_malloc(max(off, 1) * 10 + 64); d_n = root.device_malloc(64)
            root.voxel_grid_from_partials_device(d_k, d_p, off, leaf, d_o, max(off, 1) * 5, d_n)
            root.synchronize()
            nv = np.empty(1, np.int32); root.memcpy_d2h(nv, d_n)
            got = np.empty(max(int(nv[0]), 1) * 5, np.int16); root.memcpy_d2h(got, d_o)
            got = got[:int(nv[0]) * 5].reshape(-1, 5)
            shard_runs += 1
            if got.shape != want.shape or (got != want).any():
                bad += 1
                print("MISMATCH (sharded partials)", shapes, flags, stride, leaf, bounds, got.shape, want.shape, flush=True)
            # the same shards into a voxel SINK (pcs_voxel_sink_*): one more context owns the workspace and the tail, the shards
            # pre-aggregate into it in a random order; two or three calls on the sink (cold, then warm), sometimes another leaf between
            with PcsContext(cfgs[:1]) as sink_ctx:
                total = sum(caps)
                d_so = sink_ctx.device_malloc(total * 10 + 64); d_sn = sink_ctx.device_malloc(64)
                up = []
                for r, c in enumerate(ctxs):
                    a, b = bounds[r], bounds[r + 1]
                    dd = [c.device_malloc(max(d.nbytes, 16)) for d in depth[a:b]]
                    dc = [c.device_malloc(max(x.nbytes, 16)) for x in color[a:b]]
                    for ptr, arr in zip(dd + dc, depth[a:b] + color[a:b]):
                        c.memcpy_h2d(ptr, arr)
                    up.append((dd, dc))
                leaves = [leaf] * int(rng.integers(2, 4))
                if rng.random() < 0.3:
                    leaves.insert(1, int(rng.choice([3, 40, 64, 900])))
                for lf in leaves:
                    sink = sink_ctx.voxel_sink_begin(total, lf)
                    sink_ctx.synchronize()
                    for r in rng.permutation(len(ctxs)):
                        ctxs[r].process_frames_voxel_into_sink_device(up[r][0], up[r][1], sink)
                    for c in ctxs:
                        c.synchronize()
                    sink_ctx.voxel_sink_finish(sink, d_so, total * 5, d_sn)
                    sink_ctx.synchronize()
                    nv = np.empty(1, np.int32); sink_ctx.memcpy_d2h(nv, d_sn)
                    got = np.empty(max(int(nv[0]), 1) * 5, np.int16); sink_ctx.memcpy_d2h(got, d_so)
                    got = got[:max(int(nv[0]), 0) * 5].reshape(-1, 5)
                    want_s = want if lf == leaf else O.voxel_grid(stitched, lf)
                    sink_runs += 1
                    if int(nv[0]) < 0 or got.shape != want_s.shape or (got != want_s).any():
                        bad += 1
                        print("MISMATCH (sink)", shapes, flags, stride, lf, bounds, int(nv[0]), want_s.shape, flush=True)
        finally:
            for c in ctxs:
                c.close()
print(f"{runs} raster->voxel calls ({patch_runs} through the square-patch reader), {shard_runs} sharded partial merges, {sink_runs} sink calls, "
      f"{bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
