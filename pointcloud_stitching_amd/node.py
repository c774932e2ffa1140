"""ctypes mirror of include/pcs_node.h (libpcs_node.so): ONE process driving several GPUs of a node.

What it replaces in the reference: the star of edge servers + central stitcher
(src/pcs-camera-optimized.cpp:715-720 -> src/pcs-multicamera-client.cpp:363-409) — here cameras sharded over the GPUs,
one grouped RCCL exchange to GPU 0. Plumbing only; the work happens in libpcs_hip / libpcs_node.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lib as _libmod
from .api import PcsError
from .types import StreamConfig, HEADER_SHORTS, POINT_SHORTS, stream_array

NODE_LIB_PATH = os.path.join(os.path.dirname(_libmod.LIB_PATH), "libpcs_node.so")

VOXEL_PARTIALS = 0      # PCS_NODE_VOXEL_PARTIALS: per-GPU pre-aggregation, partials exchanged (default)
VOXEL_PAYLOADS = 1      # PCS_NODE_VOXEL_PAYLOADS: packed payloads gathered, voxel grid of the stitched cloud on the root


NO_EXCHANGE = 1         # PCS_NODE_NO_EXCHANGE
DIRECT_STORE = 2        # PCS_NODE_DIRECT_STORE


class NodeStats(C.Structure):
    _fields_ = [("ticket", C.c_int32), ("kernels_ms", C.c_float), ("exchange_ms", C.c_float), ("root_ms", C.c_float),
                ("exchanged_bytes", C.c_int64), ("reduced", C.c_int64), ("direct_bytes", C.c_int64),
                ("submit_host_ms", C.c_float), ("exchange_host_ms", C.c_float)]


class NodeLink(C.Structure):
    _fields_ = [("device", C.c_int32), ("root_device", C.c_int32), ("same_device", C.c_int32), ("can_access_root", C.c_int32),
                ("link_type", C.c_int32), ("hops", C.c_int32), ("performance_rank", C.c_int32), ("native_atomics", C.c_int32)]


LINK_TYPES = {-1: "unknown", 0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}


class VoxelStats(C.Structure):
    _fields_ = [("kernels_ms", C.c_float), ("exchange_ms", C.c_float), ("root_voxel_ms", C.c_float),
                ("exchanged_bytes", C.c_int64), ("partials", C.c_int32), ("voxels", C.c_int32)]


_P, _VP = C.POINTER, C.c_void_p
SYMBOLS = [
    ("pcs_node_create", C.c_int, [_P(_VP), C.c_int, _P(C.c_int), C.c_int, _P(StreamConfig), C.c_uint32, C.c_int]),
    ("pcs_node_create_ex", C.c_int, [_P(_VP), C.c_int, _P(C.c_int), C.c_int, _P(StreamConfig), C.c_uint32, C.c_int, C.c_uint32]),
    ("pcs_node_destroy", None, [_VP]),
    ("pcs_node_last_error", C.c_char_p, [_VP]),
    ("pcs_node_devices", C.c_int, [_VP]),
    ("pcs_node_rccl_ranks", C.c_int, [_VP]),
    ("pcs_node_max_payload_shorts", C.c_size_t, [_VP]),
    ("pcs_node_process", C.c_int, [_VP, _P(_VP), _P(_VP), _VP, C.c_size_t, C.c_int, _P(C.c_int), _P(C.c_int)]),
    ("pcs_node_process_device", C.c_int, [_VP, _P(_VP), _P(_VP), _VP, C.c_size_t, _P(C.c_int), _P(C.c_int)]),
    ("pcs_node_submit_device", C.c_int, [_VP, _P(_VP), _P(_VP), _VP, C.c_size_t, _P(C.c_int)]),
    ("pcs_node_wait", C.c_int, [_VP, C.c_int, _P(C.c_int), _P(C.c_int)]),
    ("pcs_node_inject_exchange_failure", C.c_int, [_VP]),
    ("pcs_node_rccl_version", C.c_int, [_VP]),
    ("pcs_node_rccl_header_version", C.c_int, []),
    ("pcs_node_rccl_library", C.c_char_p, [_VP]),
    ("pcs_node_link_info", C.c_int, [_VP, C.c_int, _P(NodeLink)]),
    ("pcs_node_probe_links", C.c_int, [_VP, C.c_size_t, C.c_int, _P(C.c_float)]),
    ("pcs_node_set_timing", C.c_int, [_VP, C.c_int]),
    ("pcs_node_last_stats", C.c_int, [_VP, _P(NodeStats)]),
    ("pcs_node_submit_voxel_device", C.c_int, [_VP, _P(_VP), _P(_VP), C.c_int, _VP, C.c_size_t, _P(C.c_int)]),
    ("pcs_node_wait_voxel", C.c_int, [_VP, C.c_int, _P(C.c_int)]),
    ("pcs_node_voxel_reruns", C.c_int, [_VP]),
    ("pcs_node_set_one_call", C.c_int, [_VP, C.c_int]),
    ("pcs_node_set_voxel_sink", C.c_int, [_VP, C.c_int]),
    ("pcs_node_voxel_sink", C.c_int, [_VP]),
    ("pcs_node_process_voxel_device", C.c_int, [_VP, _P(_VP), _P(_VP), C.c_int, C.c_int, _VP, C.c_size_t, _P(C.c_int), _P(VoxelStats)]),
    ("pcs_node_process_voxel", C.c_int, [_VP, _P(_VP), _P(_VP), C.c_int, C.c_int, _VP, C.c_size_t, C.c_int, _P(C.c_int), _P(VoxelStats)]),
]

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    _libmod.load()                       # builds both libraries if missing; libpcs_node links libpcs_hip by rpath
    lib = C.CDLL(NODE_LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    _lib = lib
    return lib


class PcsNode:
    """streams: all cameras in global camera order; camera g belongs to devices[g // streams_per_device].
    A device id may repeat (virtual peers of one GPU: their transfers become RCCL self send/recv pairs)."""

    def __init__(self, streams: Sequence[StreamConfig], devices: Sequence[int] = (0,), flags: int = 0, downsample: int = 1,
                 node_flags: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        self.streams = list(streams)
        self.devices = list(devices)
        if not self.devices or len(self.streams) % len(self.devices):
            raise ValueError("the streams must divide evenly over the devices")
        self.per_device = len(self.streams) // len(self.devices)
        self._arr = stream_array(self.streams)
        ids = (C.c_int * len(self.devices))(*self.devices)
        rc = self._lib.pcs_node_create_ex(C.byref(self._h), len(self.devices), ids, self.per_device,
                                          C.cast(self._arr, C.POINTER(StreamConfig)), int(flags), int(downsample), int(node_flags))
        if rc != 0:
            d = self._lib.pcs_node_last_error(None)
            self._h = C.c_void_p()
            raise PcsError(rc, d.decode() if d else "")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pcs_node_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != 0:
            d = self._lib.pcs_node_last_error(self._h)
            raise PcsError(rc, d.decode() if d else "")

    @property
    def max_payload_shorts(self) -> int:
        return int(self._lib.pcs_node_max_payload_shorts(self._h))

    @property
    def rccl_ranks(self) -> int:
        return int(self._lib.pcs_node_rccl_ranks(self._h))

    @property
    def rccl_version(self) -> int:
        """ncclGetVersion() of the librccl the process bound (0: the node has no communicator)."""
        return int(self._lib.pcs_node_rccl_version(self._h))

    @property
    def rccl_header_version(self) -> int:
        return int(self._lib.pcs_node_rccl_header_version())

    @property
    def rccl_library(self) -> str:
        d = self._lib.pcs_node_rccl_library(self._h)
        return d.decode() if d else ""

    def link_info(self, peer: int) -> dict:
        ln = NodeLink()
        self._check(self._lib.pcs_node_link_info(self._h, int(peer), C.byref(ln)))
        d = {f: int(getattr(ln, f)) for f, _ in NodeLink._fields_}
        d["link"] = "same GPU" if d["same_device"] else LINK_TYPES.get(d["link_type"], str(d["link_type"]))
        return d

    def probe_links(self, nbytes: int = 0, repeats: int = 5) -> List[float]:
        """ms per peer for `nbytes` into the root, one ncclSend/ncclRecv pair at a time (entry 0, the root, is 0)."""
        ms = (C.c_float * len(self.devices))()
        self._check(self._lib.pcs_node_probe_links(self._h, int(nbytes), int(repeats), ms))
        return [float(x) for x in ms]

    def inject_exchange_failure(self) -> None:
        self._check(self._lib.pcs_node_inject_exchange_failure(self._h))

    def set_timing(self, enable: bool) -> None:
        self._check(self._lib.pcs_node_set_timing(self._h, int(bool(enable))))

    def last_stats(self) -> dict:
        st = NodeStats()
        self._check(self._lib.pcs_node_last_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in NodeStats._fields_}

    def _raster_ptrs(self, depth, color):
        n = len(self.streams)
        if len(depth) != n or len(color) != n:
            raise ValueError("need one depth and one colour raster per stream")
        d = [np.ascontiguousarray(x, np.uint16).reshape(-1) for x in depth]
        c = [np.ascontiguousarray(x, np.uint8).reshape(-1) for x in color]
        keep = (d, c)
        return (C.c_void_p * n)(*[x.ctypes.data for x in d]), (C.c_void_p * n)(*[x.ctypes.data for x in c]), keep

    def process(self, depth, color, write_header: bool = True) -> Tuple[np.ndarray, List[int], int]:
        """pcs_node_process: host rasters in, stitched buffer (2 header shorts + records) out."""
        dp, cp, _keep = self._raster_ptrs(depth, color)
        buf = np.zeros(HEADER_SHORTS + self.max_payload_shorts, np.int16)
        counts = (C.c_int * len(self.streams))()
        size = C.c_int(0)
        self._check(self._lib.pcs_node_process(self._h, dp, cp, buf.ctypes.data, buf.size, int(write_header), counts, C.byref(size)))
        return buf, [int(x) for x in counts], size.value

    def process_device(self, d_depth: Sequence[int], d_color: Sequence[int], d_stitched: int, stitched_shorts: int):
        n = len(self.streams)
        counts = (C.c_int * n)()
        total = C.c_int(0)
        self._check(self._lib.pcs_node_process_device(self._h, (C.c_void_p * n)(*d_depth), (C.c_void_p * n)(*d_color), d_stitched,
                                                      stitched_shorts, counts, C.byref(total)))
        return [int(x) for x in counts], total.value

    def submit_device(self, d_depth: Sequence[int], d_color: Sequence[int], d_stitched: int, stitched_shorts: int) -> int:
        n = len(self.streams)
        t = C.c_int(-1)
        self._check(self._lib.pcs_node_submit_device(self._h, (C.c_void_p * n)(*d_depth), (C.c_void_p * n)(*d_color), d_stitched,
                                                     stitched_shorts, C.byref(t)))
        return t.value

    def wait(self, ticket: int):
        counts = (C.c_int * len(self.streams))()
        total = C.c_int(0)
        self._check(self._lib.pcs_node_wait(self._h, int(ticket), counts, C.byref(total)))
        return [int(x) for x in counts], total.value

    def submit_voxel_device(self, d_depth: Sequence[int], d_color: Sequence[int], leaf_mm: int, d_voxels: int, voxels_shorts: int) -> int:
        n = len(self.streams)
        t = C.c_int(-1)
        self._check(self._lib.pcs_node_submit_voxel_device(self._h, (C.c_void_p * n)(*d_depth), (C.c_void_p * n)(*d_color), int(leaf_mm),
                                                           d_voxels, voxels_shorts, C.byref(t)))
        return t.value

    def wait_voxel(self, ticket: int) -> int:
        nv = C.c_int(0)
        self._check(self._lib.pcs_node_wait_voxel(self._h, int(ticket), C.byref(nv)))
        return nv.value

    def set_one_call(self, mode: int) -> None:
        """A one-peer node: 2 (default) rasters -> voxels enqueued at submit on two contexts used in turn, 1 on one context, 0 the
        partials pipeline of a node of several peers. pcs_node_set_one_call; nothing may be in flight."""
        self._check(self._lib.pcs_node_set_one_call(self._h, int(mode)))

    def set_voxel_sink(self, on: bool) -> None:
        """Several peers that all share one GPU: True (default) every peer pre-aggregates into a sink context of that GPU and nothing is
        exchanged, False the partials exchange (RCCL self send/recv). pcs_node_set_voxel_sink; nothing may be in flight."""
        self._check(self._lib.pcs_node_set_voxel_sink(self._h, 1 if on else 0))

    @property
    def voxel_sink(self) -> bool:
        """Will the next voxel ticket go through a sink (pcs_node_voxel_sink)?"""
        return bool(self._lib.pcs_node_voxel_sink(self._h))

    def voxel_reruns(self) -> int:
        """Voxel frame-sets this node ran again on the LSD tail after a flagged bucket tail (pcs_node_voxel_reruns)."""
        return int(self._lib.pcs_node_voxel_reruns(self._h))

    def process_voxel(self, depth, color, leaf_mm: int, route: int = VOXEL_PARTIALS):
        """pcs_node_process_voxel: host rasters in -> (voxel records int16 [n,5], stats dict)."""
        dp, cp, _keep = self._raster_ptrs(depth, color)
        buf = np.zeros(HEADER_SHORTS + self.max_payload_shorts, np.int16)
        size = C.c_int(0)
        st = VoxelStats()
        self._check(self._lib.pcs_node_process_voxel(self._h, dp, cp, int(leaf_mm), int(route), buf.ctypes.data, buf.size, 1,
                                                     C.byref(size), C.byref(st)))
        n = size.value // (2 * POINT_SHORTS)
        stats = {f: getattr(st, f) for f, _ in VoxelStats._fields_}
        return buf[HEADER_SHORTS:HEADER_SHORTS + n * POINT_SHORTS].reshape(-1, POINT_SHORTS).copy(), stats

    def process_voxel_device(self, d_depth: Sequence[int], d_color: Sequence[int], leaf_mm: int, d_voxels: int,
                             voxels_shorts: int, route: int = VOXEL_PARTIALS):
        n = len(self.streams)
        nv = C.c_int(0)
        st = VoxelStats()
        self._check(self._lib.pcs_node_process_voxel_device(self._h, (C.c_void_p * n)(*d_depth), (C.c_void_p * n)(*d_color),
                                                            int(leaf_mm), int(route), d_voxels, voxels_shorts, C.byref(nv), C.byref(st)))
        return nv.value, {f: getattr(st, f) for f, _ in VoxelStats._fields_}
