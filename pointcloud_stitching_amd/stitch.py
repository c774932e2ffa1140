"""Multi-GPU stitch: camera streams are sharded across ranks (one process per GPU); rank 0 ends up
with the a7 concatenation in global camera order.

Replaces the reference's star of blocking TCP pulls (src/pcs-camera-optimized.cpp:715-720 on the edge,
readCloud + sendStitchToUnity on the centre, src/pcs-multicamera-client.cpp:363-409) with one gather
over RCCL/xGMI (torch.distributed backend "nccl"; "gloo" on CPU for the tests).

Global camera index = rank * streams_per_rank + local stream index, so concatenating the ranks'
payloads in rank order IS the reference's camera-order concatenation. With compaction the per-rank
point counts differ: the counts are all-gathered ON THE DEVICE (one collective, one device->host copy of
`world` integers into a page-locked mirror — no per-rank `.item()`), then payloads travel as point-to-point
sends into the right offset of the root buffer.

BASELINE configs[4] (16 x 1920x1080, 2 per GPU, compaction + voxel grid of the stitched cloud) is
`ShardedVoxelGrid`: every rank pre-aggregates its own cameras into voxel partials (integer sums, so the
grid of the union is the grid of the stitched cloud), the partials — not the points — are gathered, and the
root runs one sort + segmented mean (pcs_voxel_grid_from_partials_device). Sequence per rank and frame-set:
  1. pcs_process_frames_voxel_partials_device on the rank's stream (root: into the head of the merged arrays)
  2. all_gather of the partial counts (device tensors) -> one copy to the host
  3. grouped isend / irecv: m_r keys (8 B) and m_r partials (32 B) to the root, behind ranks < r
  4. root: pcs_voxel_grid_from_partials_device over sum(m_r) partials
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch
import torch.distributed as dist

from .types import POINT_SHORTS

KEY_BYTES, PARTIAL_BYTES = 8, 32        # pcs_voxel_partial wire format (include/pcs_hip.h)


def _as_bytes(t: torch.Tensor) -> torch.Tensor:
    """The records are opaque 10-byte units on the wire; uint8 is the one dtype every backend moves."""
    return t.view(torch.uint8)


class _Completed:
    """What an exchange that needed no communication returns in place of a work handle."""

    def wait(self):
        return True

    def is_completed(self):
        return True


class RankStitcher:
    def __init__(self, group=None, root: int = 0):
        self.group = group
        self.root = root
        self._h_counts: Optional[torch.Tensor] = None
        if not dist.is_initialized():          # one GPU, no process group: every exchange is the identity
            self.rank, self.world, self.host_staged = 0, 1, False
            return
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # gloo moves host memory only: device tensors are staged through the host then (the CPU tests, and the
        # two-ranks-on-one-GPU control-flow tests of the GPU tier; RCCL — backend "nccl" — moves device memory itself)
        self.host_staged = dist.get_backend(group) == "gloo"

    # fixed-size case: every rank contributes exactly `points_per_rank` points
    def gather_fixed(self, local_payload: torch.Tensor, stitched: Optional[torch.Tensor], async_op: bool = False):
        """local_payload: int16 [points*5]; stitched (root only): int16 [world*points*5]."""
        n = local_payload.numel()
        src = _as_bytes(local_payload)
        if self.world == 1:                     # no process group needed: the gather of one rank is a copy (or nothing at all)
            if stitched is None or stitched.numel() < n:
                raise ValueError("root needs a stitched buffer of world * local size")
            if stitched.data_ptr() != local_payload.data_ptr():
                _as_bytes(stitched[:n]).copy_(src)
            return _Completed()
        if self.rank == self.root:
            if stitched is None or stitched.numel() < n * self.world:
                raise ValueError("root needs a stitched buffer of world * local size")
            views = [_as_bytes(stitched[r * n:(r + 1) * n]) for r in range(self.world)]
            return dist.gather(src, views, dst=self.root, group=self.group, async_op=async_op)
        return dist.gather(src, None, dst=self.root, group=self.group, async_op=async_op)

    # variable-size case (after compaction)
    def gather_counts(self, local_count: Union[int, torch.Tensor], device) -> List[int]:
        """Every rank's count. `local_count` may be a host int or a 1-element integer tensor that already lives on the
        device (e.g. a view of the counts word the compaction kernel wrote): then nothing is read back before the
        collective. One all_gather on the device, one copy of `world` integers to a page-locked host mirror."""
        device = torch.device(device)
        if isinstance(local_count, torch.Tensor):
            t = local_count.reshape(1).to(device=device, dtype=torch.int64)
        else:
            t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
        if self.world == 1:
            return [int(t.item())]
        if self.host_staged and device.type == "cuda":
            t, device = t.cpu(), torch.device("cpu")
        out = torch.empty(self.world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        if device.type == "cuda":
            if self._h_counts is None:
                self._h_counts = torch.empty(self.world, dtype=torch.int64).pin_memory()
            self._h_counts.copy_(out, non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
            return self._h_counts.tolist()
        return out.tolist()

    def gather_bytes(self, local: Sequence[torch.Tensor], sizes: Sequence[Sequence[int]],
                     merged: Optional[Sequence[torch.Tensor]]) -> None:
        """ONE grouped exchange of several variable-length byte arrays per rank. local[a]: uint8 tensor of this rank
        (its first sizes[a][rank] bytes travel); merged[a] (root): uint8 tensor that receives rank r's bytes at offset
        sum(sizes[a][:r]). The root's own bytes are copied unless they already sit there (same storage address)."""
        ops, own, landed = [], [], []
        stage = self.host_staged
        for a, src in enumerate(local):
            sz = sizes[a]
            if self.rank == self.root:
                off = 0
                for r in range(self.world):
                    if sz[r]:
                        dst = merged[a][off:off + sz[r]]
                        if r == self.root:
                            if dst.data_ptr() != src.data_ptr():
                                mine = src[:sz[r]]
                                # root != 0: the root's own bytes sit at the head of the merged array and move up by `off`;
                                # when the two ranges overlap copy_ is not a memmove, and the ranks below the root are about
                                # to land in [0, off) — go through a temporary
                                lo, hi = sorted((dst.data_ptr(), mine.data_ptr()))
                                if dst.untyped_storage().data_ptr() == mine.untyped_storage().data_ptr() and hi - lo < sz[r]:
                                    mine = mine.clone()
                                own.append((dst, mine))
                        elif stage and dst.is_cuda:
                            tmp = torch.empty(sz[r], dtype=torch.uint8)
                            landed.append((dst, tmp))
                            ops.append(dist.P2POp(dist.irecv, tmp, r, self.group))
                        else:
                            ops.append(dist.P2POp(dist.irecv, dst, r, self.group))
                    off += sz[r]
            elif sz[self.rank]:
                out = src[:sz[self.rank]]
                ops.append(dist.P2POp(dist.isend, out.cpu() if (stage and out.is_cuda) else out, self.root, self.group))
        for dst, src in own:
            dst.copy_(src)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for dst, tmp in landed:
            dst.copy_(tmp)

    def gather_variable(self, local_payload: torch.Tensor, local_points: Union[int, torch.Tensor],
                        stitched: Optional[torch.Tensor]) -> List[int]:
        """Returns every rank's point count; on root, stitched[: sum*5] holds the concatenation."""
        counts = self.gather_counts(local_points, local_payload.device)
        total = sum(counts) * POINT_SHORTS
        if self.rank == self.root and (stitched is None or stitched.numel() < total):
            raise ValueError("stitched buffer too small")
        self.gather_bytes([_as_bytes(local_payload)], [[c * POINT_SHORTS * 2 for c in counts]],
                          [_as_bytes(stitched)] if self.rank == self.root else None)
        return counts


class ShardedVoxelGrid:
    """BASELINE configs[4] with one process per GPU: `ctx` holds this rank's cameras (global camera order = rank order).
    All buffers are torch tensors on the rank's device; the root's key / partial arrays take every rank's partials."""

    def __init__(self, ctx, leaf_mm: int, device, group=None, root: int = 0, capacity_per_rank: Optional[int] = None):
        self.ctx, self.leaf = ctx, int(leaf_mm)
        self.st = RankStitcher(group, root)
        self.device = torch.device(device)
        cap = int(capacity_per_rank) if capacity_per_rank else ctx.max_payload_shorts // POINT_SHORTS
        self.cap = cap
        total = cap * (self.st.world if self.st.rank == root else 1)
        self.keys = torch.empty(total * KEY_BYTES + 64, dtype=torch.uint8, device=self.device)
        self.parts = torch.empty(total * PARTIAL_BYTES + 64, dtype=torch.uint8, device=self.device)
        self.n_local = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.n_vox = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.total_cap = total
        self.counts: List[int] = []

    # The context's kernels run on the CONTEXT's HIP stream (its own non-blocking stream unless pcs_set_stream adopted one);
    # the count read-back, the collectives and their copies run on torch's current stream. Unless the two are the same
    # stream, each side waits for the other through events — a caller that never called ctx.set_stream gets the same bytes.
    def _ctx_stream(self):
        if self.device.type != "cuda":
            return None
        h = self.ctx.get_stream()
        if not h or h == torch.cuda.current_stream(self.device).cuda_stream:
            return None
        return torch.cuda.ExternalStream(h, device=self.device)

    def _torch_waits_for_ctx(self):
        ext = self._ctx_stream()
        if ext is not None:
            torch.cuda.current_stream(self.device).wait_stream(ext)

    def _ctx_waits_for_torch(self):
        ext = self._ctx_stream()
        if ext is not None:
            ext.wait_stream(torch.cuda.current_stream(self.device))

    def pre_aggregate(self, d_depth: Sequence[int], d_color: Sequence[int]) -> None:
        """Step 1 (asynchronous on the context's stream)."""
        self.ctx.process_frames_voxel_partials_device(d_depth, d_color, self.leaf, self.keys.data_ptr(), self.parts.data_ptr(),
                                                      self.cap, self.n_local.data_ptr())

    def exchange(self) -> List[int]:
        """Steps 2 + 3. Returns every rank's partial count."""
        self._torch_waits_for_ctx()                   # the pre-aggregation wrote n_local / keys / parts on the context's stream
        self.counts = self.st.gather_counts(self.n_local[0], self.device)
        if max(self.counts) > self.cap:
            raise RuntimeError(f"a rank reported {max(self.counts)} partials (capacity {self.cap})")
        self.st.gather_bytes([self.keys, self.parts],
                             [[c * KEY_BYTES for c in self.counts], [c * PARTIAL_BYTES for c in self.counts]],
                             [self.keys, self.parts] if self.st.rank == self.st.root else None)
        return self.counts

    def reduce(self, d_out: int, out_shorts: int) -> None:
        """Step 4, root only (asynchronous; the voxel count lands in self.n_vox[0])."""
        if self.st.rank != self.st.root:
            return
        self._ctx_waits_for_torch()                   # the received partials landed behind torch's current stream
        self.ctx.voxel_grid_from_partials_device(self.keys.data_ptr(), self.parts.data_ptr(), sum(self.counts), self.leaf,
                                                 d_out, out_shorts, self.n_vox.data_ptr())

    def voxels(self, d_out: int, out_shorts: int) -> int:
        """Root: the voxel count of the last reduce(), read back (synchronises the context's stream) — never negative. A bucket tail
        that gave up waiting for one of its own workgroups leaves -1 there (include/pcs_hip.h, pcs_voxel_grid_device): the partials
        are still in the root's arrays, so the reduce is run again on the LSD tail, which is then latched for the context. A negative
        length must never reach a caller's size arithmetic or the wire (src/pcs-multicamera-client.cpp:394-403)."""
        if self.st.rank != self.st.root:
            return 0
        self.ctx.synchronize()
        nv = int(self.n_vox[0].item())
        if nv < 0:
            self.ctx.set_voxel_tail(3)                # PCS_VOXEL_TAIL_LSD_LATCHED
            self.reduce(d_out, out_shorts)
            self.ctx.synchronize()
            nv = int(self.n_vox[0].item())
            if nv < 0:
                raise RuntimeError("the voxel pipeline reported a negative count on the LSD tail too (device stalled?)")
        return nv

    def run(self, d_depth: Sequence[int], d_color: Sequence[int], d_out: int, out_shorts: int) -> None:
        self.pre_aggregate(d_depth, d_color)
        self.exchange()
        self.reduce(d_out, out_shorts)
