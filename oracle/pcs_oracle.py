"""ctypes front-end of oracle/libpcs_oracle.so.  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
anything under pointcloud_stitching_amd/. See oracle/pcs_oracle.h for the parity-pin status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

from pointcloud_stitching_amd.types import (POINT_SHORTS, HEADER_SHORTS, StreamConfig, stream_array)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpcs_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Always runs make (a no-op when the library is newer than every source and header it depends on, incl.
    include/pcs_hip.h): a checker left over from before a header change must not survive it."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:      # two test processes must not run make at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout[-4000:])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        SC = P(StreamConfig)
        L.pcs_oracle_cvtt.restype = C.c_int32
        L.pcs_oracle_cvtt.argtypes = [C.c_float]
        L.pcs_oracle_deproject.restype = None
        L.pcs_oracle_deproject.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pcs_oracle_pack.restype = C.c_int
        L.pcs_oracle_pack.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.pcs_oracle_pack_scalar_variant.restype = C.c_int
        L.pcs_oracle_pack_scalar_variant.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pcs_oracle_send_xyzrgb_pointcloud.restype = C.c_int
        L.pcs_oracle_send_xyzrgb_pointcloud.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32,
                                                         C.c_void_p, C.c_size_t, C.c_int]
        L.pcs_oracle_stitch.restype = C.c_int
        L.pcs_oracle_stitch.argtypes = [P(C.c_void_p), P(C.c_int), C.c_int, C.c_int, C.c_void_p]
        L.pcs_oracle_process_frames.restype = C.c_int
        L.pcs_oracle_process_frames.argtypes = [SC, C.c_int, P(C.c_void_p), P(C.c_void_p), C.c_uint32, C.c_int,
                                                 C.c_void_p, P(C.c_int)]
        L.pcs_oracle_voxel_grid.restype = C.c_int
        L.pcs_oracle_voxel_grid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pcs_oracle_simd_available.restype = C.c_int
        L.pcs_oracle_max_threads.restype = C.c_int
        L.pcs_oracle_pack_simd_omp.restype = C.c_int
        L.pcs_oracle_pack_simd_omp.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.pcs_oracle_deproject_omp.restype = None
        L.pcs_oracle_deproject_omp.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pcs_oracle_send_simd_omp.restype = C.c_int
        L.pcs_oracle_send_simd_omp.argtypes = [SC, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.pcs_oracle_transform_payload.restype = C.c_int
        L.pcs_oracle_transform_payload.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.pcs_oracle_place_omp.restype = None
        L.pcs_oracle_place_omp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.pcs_oracle_team_cpus.restype = None
        L.pcs_oracle_team_cpus.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _p(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def cvtt(f: float) -> int:
    return int(lib().pcs_oracle_cvtt(C.c_float(f)))


def deproject(sc: StreamConfig, depth: np.ndarray, flags: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    depth = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
    n = sc.n_points
    assert depth.size == n
    vtx = np.empty((n, 3), np.float32)
    tex = np.empty((n, 2), np.float32)
    L = lib()
    L.pcs_oracle_deproject_flags.restype = None
    L.pcs_oracle_deproject_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.pcs_oracle_deproject_flags(C.byref(sc), _p(depth), flags, _p(vtx), _p(tex))
    return vtx, tex


def pack(sc: StreamConfig, vertices: np.ndarray, texcoords: np.ndarray, color: np.ndarray,
         flags: int = 0, downsample: int = 1) -> np.ndarray:
    vtx = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    tex = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    col = np.ascontiguousarray(color, np.uint8).reshape(-1)
    n = vtx.shape[0]
    out = np.zeros((max(n, 1), POINT_SHORTS), np.int16)
    cnt = lib().pcs_oracle_pack(C.byref(sc), _p(vtx), _p(tex), n, _p(col), flags, downsample, _p(out))
    return out[:cnt].copy()


def pack_scalar_variant(sc, vertices, texcoords, color) -> np.ndarray:
    vtx = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    tex = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    col = np.ascontiguousarray(color, np.uint8).reshape(-1)
    out = np.zeros((max(vtx.shape[0], 1), POINT_SHORTS), np.int16)
    cnt = lib().pcs_oracle_pack_scalar_variant(C.byref(sc), _p(vtx), _p(tex), vtx.shape[0], _p(col), _p(out))
    return out[:cnt].copy()


def pack_simd_omp(sc, vertices, texcoords, color, n_threads: int = 1) -> np.ndarray:
    vtx = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    tex = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    col = np.ascontiguousarray(color, np.uint8).reshape(-1)
    out = np.zeros((max(vtx.shape[0], 1), POINT_SHORTS), np.int16)
    cnt = lib().pcs_oracle_pack_simd_omp(C.byref(sc), _p(vtx), _p(tex), vtx.shape[0], _p(col), _p(out), n_threads)
    return out[:cnt].copy()


def deproject_omp(sc, depth, n_threads: int = 1):
    depth = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
    n = sc.n_points
    vtx = np.empty((n, 3), np.float32)
    tex = np.empty((n, 2), np.float32)
    lib().pcs_oracle_deproject_omp(C.byref(sc), _p(depth), _p(vtx), _p(tex), n_threads)
    return vtx, tex


def send_xyzrgb_pointcloud(sc, vertices, texcoords, color, flags: int = 0,
                           buffer_shorts: int = 5_000_000, write_header: bool = True,
                           prefill: int = 0x5A5A) -> Tuple[np.ndarray, int]:
    vtx = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    tex = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    col = np.ascontiguousarray(color, np.uint8).reshape(-1)
    buf = np.full(buffer_shorts, prefill, np.uint16).view(np.int16)
    size = lib().pcs_oracle_send_xyzrgb_pointcloud(C.byref(sc), _p(vtx), _p(tex), vtx.shape[0], _p(col), flags,
                                                   _p(buf), buffer_shorts, int(write_header))
    return buf, int(size)


def stitch(cam_payloads: Sequence[np.ndarray], downsample: int = 1) -> np.ndarray:
    cams = [np.ascontiguousarray(p, np.int16).reshape(-1, POINT_SHORTS) for p in cam_payloads]
    n = len(cams)
    ptrs = (C.c_void_p * n)(*[_p(c) if c.size else None for c in cams])
    cnts = (C.c_int * n)(*[c.shape[0] for c in cams])
    total = sum(-(-c.shape[0] // max(downsample, 1)) for c in cams)
    out = np.zeros((max(total, 1), POINT_SHORTS), np.int16)
    got = lib().pcs_oracle_stitch(ptrs, cnts, n, downsample, _p(out))
    assert got == total
    return out[:got].copy()


def process_frames(configs: Sequence[StreamConfig], depth: Sequence[np.ndarray], color: Sequence[np.ndarray],
                   flags: int = 0, downsample: int = 1) -> Tuple[np.ndarray, List[int]]:
    n = len(configs)
    arr = stream_array(configs)
    d = [np.ascontiguousarray(x, np.uint16).reshape(-1) for x in depth]
    c = [np.ascontiguousarray(x, np.uint8).reshape(-1) for x in color]
    dp = (C.c_void_p * n)(*[_p(x) for x in d])
    cp = (C.c_void_p * n)(*[_p(x) for x in c])
    total_max = sum(cfg.n_points for cfg in configs)
    out = np.zeros((max(total_max, 1), POINT_SHORTS), np.int16)
    counts = (C.c_int * n)()
    tot = lib().pcs_oracle_process_frames(arr, n, dp, cp, flags, downsample, _p(out), counts)
    if tot < 0:
        raise MemoryError("oracle out of memory")
    return out[:tot].copy(), [int(x) for x in counts]


def voxel_grid(payload: np.ndarray, leaf_mm: int) -> np.ndarray:
    p = np.ascontiguousarray(payload, np.int16).reshape(-1, POINT_SHORTS)
    out = np.zeros((max(p.shape[0], 1), POINT_SHORTS), np.int16)
    nv = lib().pcs_oracle_voxel_grid(_p(p), p.shape[0], int(leaf_mm), _p(out))
    if nv < 0:
        raise MemoryError
    return out[:nv].copy()


def transform_payload(payload: np.ndarray, m16, downsample: int = 1) -> np.ndarray:
    """pcs-multicamera-optimized's centre-side decode / pcl::transformPointCloud / re-encode of one camera's records."""
    p = np.ascontiguousarray(payload, np.int16).reshape(-1, POINT_SHORTS)
    m = np.ascontiguousarray(np.asarray(m16, np.float32).reshape(-1))
    assert m.size == 16
    out = np.empty_like(p)
    n = lib().pcs_oracle_transform_payload(_p(p), p.shape[0], int(downsample), _p(m), _p(out))
    return out[:n].copy()
