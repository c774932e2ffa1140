"""Deterministic synthetic camera frames — no files, no libm.

The reference's inputs (samples/*.bag) are Git-LFS stubs and librealsense is absent, so every
workload is generated. Integer-only arithmetic (a counter-based 32-bit hash and a triangle wave)
makes the frames identical on every host and in the C++ CLI's generator (csrc/pcs_synth.h).

  depth  Z16  : smooth scene 500..4500 mm (two-axis triangle wave, phase shifted per stream)
                + uniform noise in [-8, 8], then ~10 % Bernoulli holes and one solid 64x64 hole -> 0
  colour RGB8 : incompressible (hash bytes), stride = 3*W
  intrinsics  : fx = fy = 0.7*W, ppx = W/2 - 0.5 + 3.7, ppy = H/2 - 0.5 - 2.1, no distortion
  depth->colour: R = I, t = (0.015, 0, 0) m; depth scale 0.001
  camera->world: stream s uses transform[s mod 8] (src/pcs-multicamera-optimized.cpp:417-455);
                 single-stream cases use tf_mat (src/pcs-camera-optimized.cpp:64-67)
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .types import (StreamConfig, TF_MAT, TRANSFORMS, make_intrinsics, make_stream_config)

SEED = 0xC0FFEE


def hash32(x: np.ndarray) -> np.ndarray:
    """murmur3 fmix32 on uint32 (wraps mod 2^32)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x85EBCA6B)
    x ^= x >> np.uint32(13)
    x *= np.uint32(0xC2B2AE35)
    x ^= x >> np.uint32(16)
    return x


def _tri1024(x: np.ndarray) -> np.ndarray:
    """Triangle wave of period 1024 with range 0..512."""
    m = x & 1023
    return np.where(m < 512, m, 1024 - m)


def synth_depth(width: int, height: int, stream: int = 0, seed: int = SEED, mode: str = "scene") -> np.ndarray:
    n = width * height
    idx = np.arange(n, dtype=np.uint32)
    key = np.uint32((seed + 0x9E3779B9 * (stream + 1)) & 0xFFFFFFFF)
    h = hash32(idx * np.uint32(2654435761) + key)
    if mode == "random":      # worst case: uniform over the whole Z16 range
        return (h & np.uint32(0xFFFF)).astype(np.uint16).reshape(height, width)
    r = (idx // np.uint32(width)).astype(np.int64)
    c = (idx % np.uint32(width)).astype(np.int64)
    phase = (3 * c * 1024) // width + (2 * r * 1024) // height + 128 * stream
    d = 500 + (_tri1024(phase) * 4000) // 512
    d = d + (h & np.uint32(0xF)).astype(np.int64) - 8 + ((h >> np.uint32(4)) & np.uint32(1)).astype(np.int64)
    hole = ((h >> np.uint32(8)) % np.uint32(10)) == 0
    bx, by = (width // 3) & ~7, (height // 4)
    block = (c >= bx) & (c < bx + 64) & (r >= by) & (r < by + 64)
    d = np.where(hole | block, 0, d)
    return d.astype(np.uint16).reshape(height, width)


def synth_color(width: int, height: int, stream: int = 0, seed: int = SEED, bpp: int = 3,
                stride: Optional[int] = None) -> np.ndarray:
    stride = stride if stride is not None else bpp * width
    nbytes = stride * height
    nwords = (nbytes + 3) // 4
    key = np.uint32((seed ^ 0x5BD1E995) + 0x7F4A7C15 * (stream + 1) & 0xFFFFFFFF)
    w = hash32(np.arange(nwords, dtype=np.uint32) * np.uint32(0x9E3779B1) + key)
    return w.view(np.uint8)[:nbytes].copy()


def default_intrinsics(width: int, height: int):
    return make_intrinsics(width, height, fx=0.7 * width, fy=0.7 * width,
                           ppx=width / 2 - 0.5 + 3.7, ppy=height / 2 - 0.5 - 2.1)


def synth_stream_config(width: int, height: int, stream: int = 0, single: bool = False,
                        color_size: Optional[Tuple[int, int]] = None) -> StreamConfig:
    di = default_intrinsics(width, height)
    ci = default_intrinsics(*color_size) if color_size else None
    m = TF_MAT if single else TRANSFORMS[stream % 8]
    return make_stream_config(di, ci, cam_to_world=m)


def synth_frame_set(n_streams: int, width: int, height: int, seed: int = SEED, single: bool = False,
                    mode: str = "scene"):
    """Returns (configs, depth list [H,W] uint16, colour list flat uint8)."""
    cfgs = [synth_stream_config(width, height, s, single=single and n_streams == 1) for s in range(n_streams)]
    depth = [synth_depth(width, height, s, seed, mode) for s in range(n_streams)]
    color = [synth_color(cfgs[s].color.width, cfgs[s].color.height, s, seed) for s in range(n_streams)]
    return cfgs, depth, color


def write_pcsraw(path: str, configs, frames) -> None:
    """Raw frame dump read by the C++ CLI (-f file.pcsraw): magic "PCSRAW1\\0", int32 n_streams,
    int32 n_frames, n_streams x pcs_stream_config, then per frame per stream: Z16 raster, colour raster.
    `frames` is a list of (depth_list, color_list)."""
    import ctypes as C
    import struct
    with open(path, "wb") as f:
        f.write(b"PCSRAW1\0")
        f.write(struct.pack("<ii", len(configs), len(frames)))
        for c in configs:
            f.write(bytes(memoryview(C.string_at(C.addressof(c), C.sizeof(c)))))
        for depth, color in frames:
            for s in range(len(configs)):
                f.write(np.ascontiguousarray(depth[s], np.uint16).tobytes())
                f.write(np.ascontiguousarray(color[s], np.uint8).tobytes())
