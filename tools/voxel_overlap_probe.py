#!/usr/bin/env python3
"""GPU box: BASELINE configs[4] in one call from the rasters (pcs_process_frames_voxel_device, 16 x 1920x1080, DROP_INVALID), frame
loop over C contexts used in turn (each its own stream, workspace, splitters and regions): with C = 2 the bucket tail of frame-set k
runs beside the pre-aggregation of k+1. Host clock over the loop, both clouds checked against the oracle digest.
   python tools/voxel_overlap_probe.py [contexts=2] [leaf=50] [reps=200]"""
import hashlib
import json
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as Syn
from pointcloud_stitching_amd.api import PcsContext

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 2
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
S, W, H = 16, 1920, 1080
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config5_digests.json")))["voxel"].get(str(leaf))
dev = torch.device("cuda", 0)
cfgs = [Syn.synth_stream_config(W, H, s) for s in range(S)]
ctxs = [PcsContext(cfgs, flags=4) for _ in range(NC)]          # own non-blocking streams ...
for c in ctxs[1:]:
    c.use_stream_beside(ctxs[0])                               # ... on different hardware queues
n = W * H
dep0 = [torch.from_numpy(Syn.synth_depth(W, H, s).reshape(-1).view(np.uint8)).to(dev) for s in range(S)]
col0 = [torch.from_numpy(Syn.synth_color(W, H, s)).to(dev) for s in range(S)]
sets = [(dep0, col0)] + [([d.clone() for d in dep0], [c.clone() for c in col0]) for _ in range(3)]
vox = [torch.empty(S * n * 5, dtype=torch.int16, device=dev) for _ in range(NC)]
nv = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(NC)]
torch.cuda.synchronize()
k = [0]
def call():
    i = k[0] % NC
    d, c = sets[k[0] % 4]; k[0] += 1
    ctxs[i].process_frames_voxel_device([t.data_ptr() for t in d], [t.data_ptr() for t in c], leaf, vox[i].data_ptr(), vox[i].numel(), nv[i].data_ptr())
def sync():
    for c in ctxs:
        c.synchronize()
for _ in range(4 * NC):
    call()
sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    sync()
    best = min(best, (time.perf_counter() - t0) / reps)
ok = []
for i in range(NC):
    m = int(nv[i][0].item())
    dig = hashlib.sha256(vox[i][:m * 5].cpu().numpy().tobytes()).hexdigest()
    ok.append(bool(gold and dig == gold["sha256"] and m == gold["voxels"]))
tag = " ".join(f"{k_}={v}" for k_, v in sorted(os.environ.items()) if k_.startswith("PCS_VOXEL"))
print(f"[{tag}] {NC} context(s), leaf {leaf} mm: {best * 1e3:.4f} ms per frame-set (host clock, best of 3 x {reps}), digests {ok}", flush=True)
for c in ctxs:
    c.close()
