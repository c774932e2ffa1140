#!/usr/bin/env python3
"""Regenerates tests/golden/synthetic_digests.json: SHA-256 of the synthetic frames and of the oracle's
packed output on them. Run from the repo root: python tests/golden/make_golden.py
(The reference cannot be executed here — see oracle/pcs_oracle.h — so these digests pin the oracle
against regressions; the reference-produced vectors are tests/golden/kat_appendix_b.json.)"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointcloud_stitching_amd import synthetic as S   # noqa: E402
from oracle import pcs_oracle as O                    # noqa: E402

frames = {}
for (w, h) in [(64, 48), (640, 480), (1280, 720)]:
    for s in (0, 1, 7):
        frames[f"depth:{w}:{h}:{s}"] = hashlib.sha256(S.synth_depth(w, h, s).tobytes()).hexdigest()
        frames[f"color:{w}:{h}:{s}"] = hashlib.sha256(S.synth_color(w, h, s).tobytes()).hexdigest()

oracle = {}
for (n, w, h, flags, ds) in [(1, 64, 48, 0, 1), (1, 640, 480, 0, 1), (1, 1280, 720, 0, 1), (8, 1280, 720, 0, 1),
                             (3, 640, 480, 4, 1), (3, 640, 480, 1, 3), (2, 640, 480, 3, 1), (2, 640, 480, 0, 5)]:
    cfgs, depth, color = S.synth_frame_set(n, w, h, single=(n == 1))
    out, counts = O.process_frames(cfgs, depth, color, flags, ds)
    oracle[f"{n}:{w}:{h}:{flags}:{ds}"] = {"sha256": hashlib.sha256(out.tobytes()).hexdigest(), "counts": counts}

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synthetic_digests.json"), "w") as f:
    json.dump({"frames": frames, "oracle": oracle}, f, indent=1, sort_keys=True)
print("wrote", len(frames), "frame digests and", len(oracle), "oracle digests")
