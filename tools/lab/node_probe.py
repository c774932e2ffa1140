#!/usr/bin/env python3
"""Lab: what one host thread spends per pipelined pcs_node submit/wait when it drives P peers (virtual peers of GPU 0 when
the box has one GPU). 8 x 1280x720 in total, 8/P cameras per peer, dense and with the invalid-depth predicate.
    python tools/lab/node_probe.py [P ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as S                     # noqa: E402
from pointcloud_stitching_amd.api import PcsContext                      # noqa: E402
from pointcloud_stitching_amd.node import PcsNode, NO_EXCHANGE           # noqa: E402
from pointcloud_stitching_amd.types import FLAG_DROP_INVALID             # noqa: E402

W, H, N = 1280, 720, 8
peers = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
cfgs = [S.synth_stream_config(W, H, s) for s in range(N)]
depth = [S.synth_depth(W, H, s) for s in range(N)]
color = [S.synth_color(W, H, s) for s in range(N)]
with PcsContext(cfgs[:1]) as mem:
    dd = [mem.device_malloc(d.nbytes) for d in depth]
    dc = [mem.device_malloc(c.nbytes) for c in color]
    for p, a in zip(dd + dc, depth + color):
        mem.memcpy_h2d(p, a)
    for P in peers:
        for flags, nf, name in ((0, 0, "dense"), (FLAG_DROP_INVALID, 0, "drop_invalid"), (0, NO_EXCHANGE, "dense/no-exchange")):
            with PcsNode(cfgs, devices=[0] * P, flags=flags, node_flags=nf) as node:
                cap = node.max_payload_shorts
                out = [mem.device_malloc(cap * 2 + 64) for _ in range(2)]
                t = node.submit_device(dd, dc, out[0], cap)
                for k in range(1, 50):
                    t2 = node.submit_device(dd, dc, out[k & 1], cap); node.wait(t); t = t2
                K = 400
                t0 = time.perf_counter(); sub = 0.0
                for k in range(K):
                    a = time.perf_counter()
                    t2 = node.submit_device(dd, dc, out[k & 1], cap)
                    sub += time.perf_counter() - a
                    node.wait(t); t = t2
                node.wait(t)
                el = time.perf_counter() - t0
                print(f"P={P} {name:18s} {el / K * 1e6:8.1f} us/frame-set  (submit {sub / K * 1e6:6.1f} us)  rccl_ranks={node.rccl_ranks}", flush=True)
                for o in out:
                    mem.device_free(o)
