#!/usr/bin/env python3
"""Randomised soak of libpcs_node on the GPU box: random numbers of (virtual) peers and cameras per peer, random camera
configurations (tests/test_gpu_parity.py::_random_config), predicate flags and strides; stitch tickets and voxel tickets
interleaved through the pipelined submit / wait pairs with two frame-sets in flight, the exchange on real RCCL (self
send/recv pairs of GPU 0's communicator) — every stitched cloud and every voxel cloud compared bit for bit with the oracle.

    python tools/node_soak.py [seconds=120] [seed=1]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_stitching_amd import synthetic as S                                      # noqa: E402
from pointcloud_stitching_amd.api import PcsContext                                      # noqa: E402
from pointcloud_stitching_amd.node import PcsNode, DIRECT_STORE                          # noqa: E402
from pointcloud_stitching_amd.types import FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID   # noqa: E402
from oracle import pcs_oracle as O                                                       # noqa: E402
from tests.test_gpu_parity import _random_config                                         # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
nodes = stitched = voxels = 0
FLAGS = [0, FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_DROP_INVALID]
while time.time() < t_end:
    peers = int(rng.integers(1, 7))
    per = int(rng.integers(1, 4))
    n = peers * per
    flags = FLAGS[int(rng.integers(0, len(FLAGS)))]
    ds = int(rng.choice([1, 1, 1, 2, 5]))
    w, h = int(rng.integers(3, 90)) * 8, int(rng.integers(5, 200))        # widths in multiples of 8: the square-patch voxel reader
    if rng.random() < 0.3:
        w = int(rng.integers(17, 400))                                     # ... and others: the row reader
    rand_cam = rng.random() < 0.4
    cw, ch = (int(rng.choice([64, 100, 192, 320])), int(rng.choice([48, 75, 108, 180]))) if rand_cam else (w, h)
    cfgs = ([_random_config(rng, w, h, cw, ch, rng.random() < 0.4) for _ in range(n)] if rand_cam
            else [S.synth_stream_config(w, h, int(rng.integers(0, 8))) for _ in range(n)])
    frames = int(rng.integers(2, 6))
    sets = []
    for f in range(frames):
        depth = [S.synth_depth(w, h, s, seed=int(rng.integers(1, 1 << 30))) for s in range(n)]
        for d in depth:
            if rng.random() < 0.5:
                d[: h // 3, :] //= 4
        color = [S.synth_color(cw, ch, s, seed=int(rng.integers(1, 1 << 30))) for s in range(n)]
        sets.append((depth, color))
    want = [O.process_frames(cfgs, d, c, flags, ds) for d, c in sets]
    leaf = int(rng.choice([7, 20, 35, 50, 120, 400, 3000]))
    nf = DIRECT_STORE if rng.random() < 0.3 else 0          # (without a predicate the pack kernels then store into the root themselves)
    with PcsNode(cfgs, devices=[0] * peers, flags=flags, downsample=ds, node_flags=nf) as node, PcsContext(cfgs[:1]) as mem:
        cap = node.max_payload_shorts
        if peers > 1 and rng.random() < 0.4:
            node.set_voxel_sink(False)                # the peers share GPU 0: sinks by default, the RCCL partials exchange for the others
        dev = []
        for depth, color in sets:
            dd = [mem.device_malloc(max(d.nbytes, 16)) for d in depth]
            dc = [mem.device_malloc(max(c.nbytes, 16)) for c in color]
            for p, a in zip(dd + dc, depth + color):
                mem.memcpy_h2d(p, a)
            dev.append((dd, dc))
        out = [mem.device_malloc(cap * 2 + 64) for _ in range(2)]
        kinds = [bool(rng.random() < 0.5) for _ in range(frames)]          # True: voxel ticket

        def submit(k):
            dd, dc = dev[k]
            return (node.submit_voxel_device(dd, dc, leaf, out[k & 1], cap) if kinds[k] else node.submit_device(dd, dc, out[k & 1], cap))

        def check(k, t):
            w_pts, w_counts = want[k]
            if kinds[k]:
                nv = node.wait_voxel(t)
                wv = O.voxel_grid(w_pts, leaf)
                got = np.empty(max(nv, 1) * 5, np.int16)
                if nv:
                    mem.memcpy_d2h(got[:nv * 5], out[k & 1])
                assert nv == wv.shape[0] and (got[:nv * 5].reshape(-1, 5) == wv).all(), (seed, nodes, k, "voxel", peers, per, flags, ds, leaf, w, h)
                return 1
            counts, total = node.wait(t)
            got = np.empty(max(total, 1) * 5, np.int16)
            if total:
                mem.memcpy_d2h(got[:total * 5], out[k & 1])
            assert counts == w_counts and total == w_pts.shape[0] and (got[:total * 5].reshape(-1, 5) == w_pts).all(), \
                (seed, nodes, k, "stitch", peers, per, flags, ds, w, h)
            return 0
        t_prev = submit(0)
        for k in range(1, frames + 1):
            t_next = submit(k) if k < frames else None
            if check(k - 1, t_prev):
                voxels += 1
            else:
                stitched += 1
            t_prev = t_next
    nodes += 1
print(f"node_soak seed {seed}: {nodes} nodes, {stitched} stitched clouds + {voxels} voxel clouds through pipelined tickets, 0 mismatches", flush=True)
