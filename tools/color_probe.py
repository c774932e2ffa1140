#!/usr/bin/env python3
"""GPU box: what limits the real-camera geometry (bench.py's color_1080p leg)? Times the dense launch (8 x 1280x720 depth, cold
ring) for the four combinations of {colour raster 1280x720 | 1920x1080} x {no colour distortion | inverse Brown-Conrady
coefficients}, all with the 1-degree depth->colour rotation.   python tools/color_probe.py [launches] [only=<combo index>]"""
import math
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as Syn
from pointcloud_stitching_amd.api import PcsContext

K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
S, W, H = 8, 1280, 720
dev = torch.device("cuda", 0)
ang = math.radians(1.0)
ax = np.array([0.3, 0.9, 0.3]); ax /= np.linalg.norm(ax)
Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
Rm = np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * Kx @ Kx
combos = [((1280, 720), False), ((1280, 720), True), ((1920, 1080), False), ((1920, 1080), True)]
for ci, (csize, dist) in enumerate(combos):
    if only >= 0 and ci != only:
        continue
    cfgs = [Syn.synth_stream_config(W, H, s, color_size=csize) for s in range(S)]
    for c in cfgs:
        for k, v in enumerate(Rm.T.reshape(-1)):
            c.depth_to_color.rotation[k] = float(v)
        if dist:
            c.color.model = 2
            for k, v in enumerate((0.12, -0.28, 0.0008, -0.0005, 0.09)):
                c.color.coeffs[k] = v
    ctx = PcsContext(cfgs)
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
    n, cb = W * H, cfgs[0].color_bytes
    R = max(4, -(-2 * (256 << 20) // (S * (n * 2 + cb))) + 2)
    dep0 = [torch.from_numpy(Syn.synth_depth(W, H, s).reshape(-1).view(np.uint8)).to(dev) for s in range(S)]
    col0 = [torch.from_numpy(Syn.synth_color(csize[0], csize[1], s)).to(dev) for s in range(S)]
    sets = [(dep0, col0)] + [([d.clone() for d in dep0], [c.clone() for c in col0]) for _ in range(R - 1)]
    outs = [torch.empty(S * n * 5 + 8, dtype=torch.int16, device=dev) for _ in range(4)]
    k = [0]
    def launch():
        d, c = sets[k[0] % R]; o = outs[k[0] % 4]; k[0] += 1
        ctx.process_frames_device([t.data_ptr() for t in d], [t.data_ptr() for t in c], o.data_ptr(), S * n * 5)
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(K):
        launch()
    ctx.timer_end()
    ms = ctx.timer_elapsed_ms() / K
    print(f"colour {csize[0]}x{csize[1]} distortion={'yes' if dist else 'no '}: {ms*1e3:6.2f} us  frac {S*n*15/(ms*1e-3)/8e12:.4f}  "
          f"math {ctx.stream_math(0)}", flush=True)
    ctx.close()
    del sets, dep0, col0
