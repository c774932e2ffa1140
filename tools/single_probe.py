#!/usr/bin/env python3
"""GPU box: back-to-back launches of pcs_process_frames_device for S device-resident 1280x720 streams over a cold ring, under the
current environment (PCS_SMALL_TILES=0 / 1: 2048- / 512-point tiles). Prints the hipEvent period; run it under tools/kstats.sh for
the kernel's own begin-to-end duration.     python tools/single_probe.py [streams=1] [launches=3000] [twin]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as Syn
from pointcloud_stitching_amd.api import PcsContext

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
twin = len(sys.argv) > 3 and sys.argv[3] == "twin"
W, H = 1280, 720
npts = W * H
dev = torch.device("cuda", 0)
cfgs = [Syn.synth_stream_config(W, H, s, single=(S == 1)) for s in range(S)]
ctx = PcsContext(cfgs)
R = -(-2 * (256 << 20) // (S * npts * (20 if twin else 5))) + 2
VP = C.c_void_p
dep = [torch.from_numpy(Syn.synth_depth(W, H, s).reshape(-1).view(np.uint8)).to(dev) for s in range(S)]
col = [torch.from_numpy(Syn.synth_color(W, H, s)).to(dev) for s in range(S)]
calls = []
if twin:
    v, t = ctx.deproject(0, Syn.synth_depth(W, H, 0))
    v0 = torch.from_numpy(v.reshape(-1).view(np.uint8)).to(dev); t0 = torch.from_numpy(t.reshape(-1).view(np.uint8)).to(dev)
outs = []
for slot in range(R):
    o = torch.empty(S * npts * 5 + 64, dtype=torch.int16, device=dev); outs.append(o)
    if twin:
        calls.append((v0.clone(), t0.clone(), col[0].clone(), o))
    else:
        d = [x.clone() for x in dep]; c = [x.clone() for x in col]
        calls.append(((VP * S)(*[x.data_ptr() for x in d]), (VP * S)(*[x.data_ptr() for x in c]), VP(o.data_ptr()), d, c))
k = [0]
lib, h = ctx._lib, ctx._h
def launch():
    a = calls[k[0] % R]; k[0] += 1
    if twin:
        rc = lib.pcs_copy_pointcloud_xyzrgb_to_buffer_device(h, 0, VP(a[0].data_ptr()), VP(a[1].data_ptr()), npts, VP(a[2].data_ptr()), VP(a[3].data_ptr()), None)
    else:
        rc = lib.pcs_process_frames_device(h, a[0], a[1], a[2], S * npts * 5, None)
    assert rc == 0, lib.pcs_last_error(h)
for _ in range(2 * R):
    launch()
ctx.synchronize()
ctx.timer_begin()
for _ in range(N):
    launch()
ctx.timer_end()
ms = ctx.timer_elapsed_ms() / N
bpp = 33 if twin else 15
print(f"[PCS_SMALL_TILES={os.environ.get('PCS_SMALL_TILES', 'auto')}] {'twin' if twin else 'fused'} {S} x {W}x{H}: {ms * 1e3:.2f} us per launch (event period), "
      f"{S * npts * bpp / (ms * 1e-3) / 1e9 / 8000:.3f} of HBM peak, ring {R}", flush=True)
ctx.close()
