"""Host-side mirror of the reference's operator surface, over the C ABI (include/pcs_hip.h).

Reference seam (src/pcs-camera-optimized.cpp):
    copyPointCloudXYZRGBToBufferSIMD(pts, color, pc_buffer)  :363  -> PcsContext.copy_pointcloud_xyzrgb_to_buffer
    sendXYZRGBPointcloud(pts, color, buffer)                 :669  -> PcsContext.send_xyzrgb_pointcloud
    rs2::pointcloud::calculate / map_to                      :288  -> PcsContext.deproject
    (edge + central collapsed)                                      -> PcsContext.process_frames
and src/pcs-multicamera-client.cpp: sendStitchToUnity :373   -> PcsContext.stitch_device

Everything here is plumbing: argument marshalling and error mapping. The arithmetic lives in the
HIP kernels (csrc/pcs_kernels.hip); there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lib as _libmod
from .types import (CloudDesc, Config, StreamConfig, STATUS_NAMES, POINT_SHORTS, POINT_BYTES, HEADER_SHORTS,
                    REF_BUF_SIZE, stream_array)


class PcsError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"{name}: {detail}" if detail else name)


def device_count() -> int:
    return int(_libmod.load().pcs_device_count())


def _ptr(a: np.ndarray) -> int:
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data


class PcsContext:
    """One context per host thread; owns device constants, staging buffers and a HIP stream."""

    def __init__(self, streams: Sequence[StreamConfig], device: int = 0, flags: int = 0, downsample: int = 1):
        self._lib = _libmod.load()
        self._h = C.c_void_p()
        self.streams = list(streams)
        self.n_streams = len(self.streams)
        self.flags = int(flags)
        self.downsample = int(downsample)
        self.device = int(device)
        self._arr = stream_array(self.streams) if self.streams else None
        cfg = Config()
        cfg.device = device
        cfg.n_streams = self.n_streams
        cfg.streams = C.cast(self._arr, C.POINTER(StreamConfig)) if self._arr is not None else None
        cfg.flags = self.flags
        cfg.downsample = self.downsample
        rc = self._lib.pcs_create(C.byref(self._h), C.byref(cfg))
        if rc != 0:
            detail = self._lib.pcs_last_error(None)
            self._h = C.c_void_p()
            raise PcsError(rc, detail.decode() if detail else "")

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            for p in getattr(self, "_pinned", []):
                self._lib.pcs_host_free(self._h, p)
            self._pinned = []
            self._lib.pcs_destroy(self._h)
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
            d = self._lib.pcs_last_error(self._h)
            raise PcsError(rc, d.decode() if d else "")

    # -- queries -----------------------------------------------------------------------------
    def stream_points(self, stream: int) -> int:
        return int(self._lib.pcs_stream_points(self._h, stream))

    def stream_math(self, stream: int) -> int:
        """0 = IEEE expansion, 1 = certified reduced-instruction arithmetic, 2 = + identity-R shortcut."""
        return int(self._lib.pcs_stream_math(self._h, stream))

    def stream_color_row_const(self, stream: int) -> bool:
        """The stream's colour row is certified independent of depth (pcs_stream_color_row_const)."""
        return bool(self._lib.pcs_stream_color_row_const(self._h, stream))

    @property
    def max_payload_shorts(self) -> int:
        return int(self._lib.pcs_max_payload_shorts(self._h))

    def set_cam_to_world(self, stream: int, m16) -> None:
        m = np.ascontiguousarray(m16, np.float32).reshape(16)
        self._check(self._lib.pcs_set_cam_to_world(self._h, stream, m.ctypes.data_as(C.POINTER(C.c_float))))

    # -- a2 twin -----------------------------------------------------------------------------
    def copy_pointcloud_xyzrgb_to_buffer(self, stream: int, vertices, texcoords, color,
                                         pc_buffer: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
        """copyPointCloudXYZRGBToBufferSIMD (:363-616): returns (payload int16[count,5], count)."""
        vtx = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        tex = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
        col = np.ascontiguousarray(color, np.uint8).reshape(-1)
        n = vtx.shape[0]
        if tex.shape[0] != n:
            raise ValueError("vertices and texcoords disagree on the point count")
        if col.size < self.streams[stream].color_bytes:
            raise ValueError("colour raster smaller than stride*height")
        out = pc_buffer if pc_buffer is not None else np.zeros((max(n, 1), POINT_SHORTS), np.int16)
        if out.dtype != np.int16 or not out.flags["C_CONTIGUOUS"] or out.size < n * POINT_SHORTS:
            # the C entry point mirrors the reference and has no capacity argument: check here
            raise ValueError(f"pc_buffer must be a C-contiguous int16 array of at least {n * POINT_SHORTS} elements")
        cnt = C.c_int(0)
        self._check(self._lib.pcs_copy_pointcloud_xyzrgb_to_buffer(
            self._h, stream, _ptr(vtx), _ptr(tex), n, _ptr(col), _ptr(out), C.byref(cnt)))
        return (out.reshape(-1, POINT_SHORTS)[:cnt.value] if pc_buffer is None else out), cnt.value

    # -- a1 twin -----------------------------------------------------------------------------
    def send_xyzrgb_pointcloud(self, stream: int, vertices, texcoords, color, buffer: np.ndarray,
                               write_header: bool = True) -> int:
        """sendXYZRGBPointcloud (:669-723) minus the socket; returns the payload size in bytes."""
        vtx = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        tex = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
        col = np.ascontiguousarray(color, np.uint8).reshape(-1)
        if buffer.dtype != np.int16 or not buffer.flags["C_CONTIGUOUS"]:
            raise ValueError("buffer must be a contiguous int16 array (the reference's short* buffer)")
        size = C.c_int(0)
        self._check(self._lib.pcs_send_xyzrgb_pointcloud(
            self._h, stream, _ptr(vtx), _ptr(tex), vtx.shape[0], _ptr(col), _ptr(buffer), buffer.size,
            int(write_header), C.byref(size)))
        return size.value

    # -- fused ---------------------------------------------------------------------------------
    def process_frames(self, depth: Sequence[np.ndarray], color: Sequence[np.ndarray],
                       write_header: bool = True, out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, List[int], int]:
        """Deproject + transform + pack every stream; returns (stitched int16 buffer incl. 2 header
        shorts, per-stream point counts, payload bytes). Pass `out` (allocated once, reused) in a frame loop:
        a fresh 74 MB buffer per call costs more in first-touch page faults than the whole GPU round trip."""
        if len(depth) != self.n_streams or len(color) != self.n_streams:
            raise ValueError("need one depth and one colour raster per stream")
        d = [np.ascontiguousarray(x, np.uint16).reshape(-1) for x in depth]
        c = [np.ascontiguousarray(x, np.uint8).reshape(-1) for x in color]
        for s in range(self.n_streams):
            if d[s].size != self.streams[s].n_points:
                raise ValueError(f"stream {s}: depth raster has {d[s].size} pixels, expected {self.streams[s].n_points}")
            if c[s].size < self.streams[s].color_bytes:
                raise ValueError(f"stream {s}: colour raster smaller than stride*height")
        dp = (C.c_void_p * self.n_streams)(*[_ptr(x) for x in d])
        cp = (C.c_void_p * self.n_streams)(*[_ptr(x) for x in c])
        buf = out if out is not None else np.zeros(HEADER_SHORTS + self.max_payload_shorts, np.int16)
        if buf.dtype != np.int16 or not buf.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a contiguous int16 array")
        counts = (C.c_int * self.n_streams)()
        size = C.c_int(0)
        self._check(self._lib.pcs_process_frames(self._h, dp, cp, _ptr(buf), buf.size, int(write_header),
                                                 counts, C.byref(size)))
        return buf, [int(x) for x in counts], size.value

    def submit_frames(self, depth: Sequence[np.ndarray], color: Sequence[np.ndarray]) -> int:
        """Queue one frame-set (uploads + kernel) and return its ticket; see pcs_submit_frames. The arrays must be
        C-contiguous uint16 / uint8 already (no conversion copy is made) and stay alive until collect_frames."""
        if len(depth) != self.n_streams or len(color) != self.n_streams:
            raise ValueError("need one depth and one colour raster per stream")
        for s in range(self.n_streams):
            d, c = depth[s], color[s]
            if d.dtype != np.uint16 or c.dtype != np.uint8 or not d.flags["C_CONTIGUOUS"] or not c.flags["C_CONTIGUOUS"]:
                raise ValueError("submit_frames wants contiguous uint16 depth and uint8 colour arrays")
            if d.size != self.streams[s].n_points or c.size < self.streams[s].color_bytes:
                raise ValueError(f"stream {s}: raster size mismatch")
        dp = (C.c_void_p * self.n_streams)(*[_ptr(x) for x in depth])
        cp = (C.c_void_p * self.n_streams)(*[_ptr(x) for x in color])
        t = C.c_int(-1)
        self._check(self._lib.pcs_submit_frames(self._h, dp, cp, C.byref(t)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[t.value] = (depth, color)          # keep the rasters alive
        return t.value

    def collect_frames(self, ticket: int, out: Optional[np.ndarray] = None,
                       write_header: bool = True) -> Tuple[np.ndarray, List[int], int]:
        buf = out if out is not None else np.zeros(HEADER_SHORTS + self.max_payload_shorts, np.int16)
        if buf.dtype != np.int16 or not buf.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a contiguous int16 array")
        counts = (C.c_int * self.n_streams)()
        size = C.c_int(0)
        try:
            self._check(self._lib.pcs_collect_frames(self._h, int(ticket), _ptr(buf), buf.size, int(write_header),
                                                     counts, C.byref(size)))
        finally:
            getattr(self, "_inflight", {}).pop(ticket, None)
        return buf, [int(x) for x in counts], size.value

    def process_frames_device(self, d_depth: Sequence[int], d_color: Sequence[int], d_payload: int,
                              payload_shorts: int, d_counts: int = 0) -> None:
        """Asynchronous launch on device pointers (ints). See pcs_process_frames_device."""
        dp = (C.c_void_p * self.n_streams)(*d_depth)
        cp = (C.c_void_p * self.n_streams)(*d_color)
        self._check(self._lib.pcs_process_frames_device(self._h, dp, cp, d_payload, payload_shorts,
                                                        d_counts or None))

    def stream_tile_base(self, stream: int) -> int:
        """Index of the stream's first tile in a per-tile array (stream == n_streams: the number of tiles in all)."""
        return int(self._lib.pcs_stream_tile_base(self._h, stream))

    def process_frames_device_counted(self, d_depth: Sequence[int], d_color: Sequence[int], d_tile_kept: int, d_payload: int,
                                      payload_shorts: int, d_counts: int = 0) -> None:
        """pcs_process_frames_device with the per-tile kept counts handed in by the producer (no count pass)."""
        dp = (C.c_void_p * self.n_streams)(*d_depth)
        cp = (C.c_void_p * self.n_streams)(*d_color)
        self._check(self._lib.pcs_process_frames_device_counted(self._h, dp, cp, d_tile_kept, d_payload, payload_shorts,
                                                                d_counts or None))

    def process_frames_device_batch(self, d_depth: Sequence[Sequence[int]], d_color: Sequence[Sequence[int]],
                                    d_payload: Sequence[int], payload_shorts: int,
                                    d_counts: Optional[Sequence[int]] = None) -> None:
        """K frame-sets per call (pcs_process_frames_device_batch): d_depth[k][s], d_color[k][s], d_payload[k]."""
        k = len(d_payload)
        if len(d_depth) != k or len(d_color) != k:
            raise ValueError("need one raster list per frame-set")
        flat_d = [p for row in d_depth for p in row]
        flat_c = [p for row in d_color for p in row]
        if len(flat_d) != k * self.n_streams or len(flat_c) != k * self.n_streams:
            raise ValueError("every frame-set needs one depth and one colour pointer per stream")
        dp = (C.c_void_p * len(flat_d))(*flat_d)
        cp = (C.c_void_p * len(flat_c))(*flat_c)
        pp = (C.c_void_p * k)(*d_payload)
        cc = (C.c_void_p * k)(*d_counts) if d_counts is not None else None
        self._check(self._lib.pcs_process_frames_device_batch(self._h, k, dp, cp, pp, payload_shorts, cc))

    def copy_pointclouds_xyzrgb_to_buffer_device(self, clouds: Sequence[Tuple[int, int, int, int, int, int]],
                                                 d_out_points: int = 0) -> None:
        """Batched a2 twin on device pointers: clouds = [(stream, n_points, d_vertices, d_texcoords, d_color, d_out)]."""
        arr = (CloudDesc * max(len(clouds), 1))()
        for i, (stream, n, v, t, col, out) in enumerate(clouds):
            arr[i].stream, arr[i].n_points = int(stream), int(n)
            arr[i].vertices, arr[i].texcoords, arr[i].color, arr[i].pc_buffer = v or None, t or None, col or None, out or None
        self._check(self._lib.pcs_copy_pointclouds_xyzrgb_to_buffer_device(self._h, len(clouds), arr, d_out_points or None))

    def deproject(self, stream: int, depth) -> Tuple[np.ndarray, np.ndarray]:
        d = np.ascontiguousarray(depth, np.uint16).reshape(-1)
        n = self.streams[stream].n_points
        if d.size != n:
            raise ValueError("depth raster size mismatch")
        vtx = np.empty((n, 3), np.float32)
        tex = np.empty((n, 2), np.float32)
        self._check(self._lib.pcs_deproject(self._h, stream, _ptr(d), _ptr(vtx), _ptr(tex)))
        return vtx, tex

    # -- a7 ------------------------------------------------------------------------------------
    def stitch_device(self, d_cam_payload: Sequence[int], cam_points: Sequence[int], downsample: int,
                      d_stitched_payload: int, stitched_shorts: int) -> int:
        n = len(d_cam_payload)
        ptrs = (C.c_void_p * n)(*d_cam_payload)
        cnts = (C.c_int * n)(*cam_points)
        total = C.c_int(0)
        self._check(self._lib.pcs_stitch_device(self._h, ptrs, cnts, n, downsample, d_stitched_payload,
                                                stitched_shorts, C.byref(total)))
        return total.value

    def transform_payloads_device(self, d_cam_payload: Sequence[int], cam_points: Sequence[int], transforms, downsample: int,
                                  d_stitched_payload: int, stitched_shorts: int):
        """pcs_transform_payloads_device: what the reference's pcs-multicamera-optimized does to every camera's payload before it
        concatenates them (decode, pcl::transformPointCloud(transform[i]), re-encode; src/pcs-multicamera-optimized.cpp:226-265,
        289). Returns (points per camera, total points)."""
        from .types import PayloadDesc
        n = len(d_cam_payload)
        descs = (PayloadDesc * max(n, 1))()
        for i in range(n):
            descs[i].d_payload = d_cam_payload[i]
            descs[i].n_points = int(cam_points[i])
            m = np.asarray(transforms[i], np.float32).reshape(-1)
            if m.size != 16:
                raise ValueError("a transform is 16 floats, row-major 4x4")
            for k in range(16):
                descs[i].transform[k] = float(m[k])
        per = (C.c_int * max(n, 1))()
        total = C.c_int(0)
        self._check(self._lib.pcs_transform_payloads_device(self._h, n, descs, int(downsample), d_stitched_payload, stitched_shorts,
                                                            per, C.byref(total)))
        return [int(per[i]) for i in range(n)], total.value

    def set_voxel_tail(self, tail: int) -> None:
        """0 = by the leaf (default), 1 = bucket tail, 2 = LSD sort + segmented mean, 3 = LSD latched after a flagged call
        (what a caller of the device forms sets when it reads a voxel count of -1, before it runs the call again); see
        pcs_set_voxel_tail."""
        self._check(self._lib.pcs_set_voxel_tail(self._h, int(tail)))

    def voxel_tail_reruns(self) -> int:
        """How often the LSD tail was latched on this context after a flagged bucket-tail call (pcs_voxel_tail_reruns)."""
        return int(self._lib.pcs_voxel_tail_reruns(self._h))

    def inject_voxel_stall(self, launches: int) -> None:
        """Fault injection: the next `launches` bucket-tail launches of this process end flagged (pcs_inject_voxel_stall)."""
        self._check(self._lib.pcs_inject_voxel_stall(int(launches)))

    # -- voxel grid (defined by this build; see include/pcs_hip.h) --------------------------------
    def voxel_grid(self, payload: np.ndarray, leaf_mm: int) -> np.ndarray:
        """Voxel-grid downsample of packed records (int16 [n,5]) -> int16 [n_voxels,5]."""
        p = np.ascontiguousarray(payload, np.int16).reshape(-1, POINT_SHORTS)
        out = np.zeros((max(p.shape[0], 1), POINT_SHORTS), np.int16)
        cnt = C.c_int(0)
        self._check(self._lib.pcs_voxel_grid(self._h, _ptr(p), p.shape[0], int(leaf_mm), _ptr(out), out.size, C.byref(cnt)))
        return out[:cnt.value].copy()

    def voxel_grid_device(self, d_payload: int, n_points: int, leaf_mm: int, d_out: int, out_shorts: int,
                          d_out_points: int = 0) -> None:
        self._check(self._lib.pcs_voxel_grid_device(self._h, d_payload, n_points, int(leaf_mm), d_out, out_shorts,
                                                    d_out_points or None))

    def voxel_grid_device_counted(self, d_payload: int, d_n_points: int, max_points: int, leaf_mm: int, d_out: int,
                                  out_shorts: int, d_out_points: int = 0) -> None:
        """The point count is read from device memory (d_n_points) when the kernels run: no host round trip after a
        compaction launch. See pcs_voxel_grid_device_counted."""
        self._check(self._lib.pcs_voxel_grid_device_counted(self._h, d_payload, d_n_points, int(max_points), int(leaf_mm),
                                                            d_out, out_shorts, d_out_points or None))

    def process_frames_voxel_device(self, d_depth: Sequence[int], d_color: Sequence[int], leaf_mm: int, d_out: int,
                                    out_shorts: int, d_out_points: int = 0) -> None:
        """Rasters -> voxel grid without the stitched cloud (pcs_process_frames_voxel_device)."""
        if len(d_depth) != self.n_streams or len(d_color) != self.n_streams:
            raise ValueError("need one depth and one colour pointer per stream")
        dp = (C.c_void_p * self.n_streams)(*d_depth)
        cp = (C.c_void_p * self.n_streams)(*d_color)
        self._check(self._lib.pcs_process_frames_voxel_device(self._h, dp, cp, int(leaf_mm), d_out, out_shorts,
                                                              d_out_points or None))

    def process_frames_voxel_partials_device(self, d_depth: Sequence[int], d_color: Sequence[int], leaf_mm: int,
                                             d_keys: int, d_partials: int, capacity: int, d_n_partials: int) -> None:
        """Rasters -> voxel partials (raw keys + sums) in caller arrays: what a rank contributes to a multi-GPU voxel grid
        (pcs_process_frames_voxel_partials_device)."""
        if len(d_depth) != self.n_streams or len(d_color) != self.n_streams:
            raise ValueError("need one depth and one colour pointer per stream")
        dp = (C.c_void_p * self.n_streams)(*d_depth)
        cp = (C.c_void_p * self.n_streams)(*d_color)
        self._check(self._lib.pcs_process_frames_voxel_partials_device(self._h, dp, cp, int(leaf_mm), d_keys, d_partials,
                                                                       int(capacity), d_n_partials))

    def voxel_grid_from_partials_device(self, d_keys: int, d_partials: int, n_partials: int, leaf_mm: int, d_out: int,
                                        out_shorts: int, d_out_points: int = 0, d_n_partials: int = 0) -> None:
        """Concatenated partials of any number of ranks -> the voxel grid (pcs_voxel_grid_from_partials_device)."""
        self._check(self._lib.pcs_voxel_grid_from_partials_device(self._h, d_keys, d_partials, int(n_partials),
                                                                  d_n_partials or None, int(leaf_mm), d_out, out_shorts,
                                                                  d_out_points or None))

    # -- voxel sink: several contexts of one device pre-aggregate into this context's workspace (pcs_voxel_sink_*) -----------------
    def voxel_sink_begin(self, capacity_points: int, leaf_mm: int):
        """Open a sink on this context's stream for `capacity_points` pixels in all; returns the opaque sink (a ctypes buffer) the
        other two calls take. The caller orders the streams (pcs_hip.h)."""
        sink = (C.c_uint64 * 24)()
        self._check(self._lib.pcs_voxel_sink_begin(self._h, int(capacity_points), int(leaf_mm), C.addressof(sink)))
        return sink

    def process_frames_voxel_into_sink_device(self, d_depth: Sequence[int], d_color: Sequence[int], sink) -> None:
        """This context's rasters under its flags -> partials in the sink's workspace, on this context's stream."""
        if len(d_depth) != self.n_streams or len(d_color) != self.n_streams:
            raise ValueError("need one depth and one colour pointer per stream")
        dp = (C.c_void_p * self.n_streams)(*d_depth)
        cp = (C.c_void_p * self.n_streams)(*d_color)
        self._check(self._lib.pcs_process_frames_voxel_into_sink_device(self._h, dp, cp, C.addressof(sink)))

    def voxel_sink_finish(self, sink, d_out: int, out_shorts: int, d_out_points: int = 0) -> None:
        """The tail over everything the sink received, on this context's stream (pcs_voxel_sink_finish)."""
        self._check(self._lib.pcs_voxel_sink_finish(self._h, C.addressof(sink), d_out, out_shorts, d_out_points or None))

    # -- plumbing ------------------------------------------------------------------------------
    def set_stream(self, hip_stream: int) -> None:
        self._check(self._lib.pcs_set_stream(self._h, hip_stream or None))

    def use_stream_beside(self, other: "PcsContext") -> bool:
        """Replace this context's own stream by one that is seen to run beside `other`'s (pcs_use_stream_beside): what two contexts
        used in turn need to overlap. True when such a stream was found."""
        rc = self._lib.pcs_use_stream_beside(self._h, other._h)
        if rc < 0:
            self._check(rc)
        return rc == 1

    def get_stream(self) -> int:
        return int(self._lib.pcs_get_stream(self._h) or 0)

    def synchronize(self) -> None:
        self._check(self._lib.pcs_synchronize(self._h))

    def timer_begin(self) -> None:
        self._check(self._lib.pcs_timer_begin(self._h))

    def timer_end(self) -> None:
        self._check(self._lib.pcs_timer_end(self._h))

    def timer_elapsed_ms(self) -> float:
        ms = C.c_float(0)
        self._check(self._lib.pcs_timer_elapsed_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def kernel_timing(self, enable: bool) -> None:
        self._check(self._lib.pcs_kernel_timing(self._h, int(enable)))

    def kernel_times_ms(self, capacity: int = 65536) -> np.ndarray:
        arr = (C.c_float * capacity)()
        n = C.c_int(0)
        self._check(self._lib.pcs_kernel_times_ms(self._h, arr, capacity, C.byref(n)))
        return np.array(arr[:min(n.value, capacity)], dtype=np.float32)

    def host_array(self, shape, dtype) -> np.ndarray:
        """A numpy array in page-locked host memory (freed with the context). Use it for the rasters and the
        stitched buffer handed to process_frames so the PCIe copies run at link speed."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self._check(self._lib.pcs_host_malloc(self._h, C.byref(p), max(n, 16)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        buf = (C.c_uint8 * max(n, 16)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def host_register(self, a: np.ndarray) -> None:
        """Page-lock an array the caller owns (pcs_host_register); the host entry points then run zero copy on it."""
        self._check(self._lib.pcs_host_register(self._h, _ptr(a), a.nbytes))

    def host_unregister(self, a: np.ndarray) -> None:
        self._check(self._lib.pcs_host_unregister(self._h, _ptr(a)))

    def device_malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.pcs_device_malloc(self._h, C.byref(p), nbytes))
        return int(p.value)

    def device_free(self, d_ptr: int) -> None:
        self._check(self._lib.pcs_device_free(self._h, d_ptr))

    def memcpy_h2d(self, d_dst: int, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src)
        self._check(self._lib.pcs_memcpy_h2d(self._h, d_dst, _ptr(src), src.nbytes))

    def memcpy_d2h(self, dst: np.ndarray, d_src: int) -> None:
        self._check(self._lib.pcs_memcpy_d2h(self._h, _ptr(dst), d_src, dst.nbytes))
