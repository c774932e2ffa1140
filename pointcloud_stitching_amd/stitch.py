"""Multi-GPU stitch: camera streams are sharded across ranks (one process per GPU); rank 0 ends up
with the a7 concatenation in global camera order.

Replaces the reference's star of blocking TCP pulls (src/pcs-camera-optimized.cpp:715-720 on the edge,
readCloud + sendStitchToUnity on the centre, src/pcs-multicamera-client.cpp:363-409) with one gather
over RCCL/xGMI (torch.distributed backend "nccl"; "gloo" on CPU for the tests).

Global camera index = rank * streams_per_rank + local stream index, so concatenating the ranks'
payloads in rank order IS the reference's camera-order concatenation. With compaction the per-rank
point counts differ: counts are all-gathered first (one int64 per rank), then payloads travel as
point-to-point sends into the right offset of the root buffer.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .types import POINT_SHORTS


def _as_bytes(t: torch.Tensor) -> torch.Tensor:
    """The records are opaque 10-byte units on the wire; uint8 is the one dtype every backend moves."""
    return t.view(torch.uint8)


class RankStitcher:
    def __init__(self, group=None, root: int = 0):
        self.group = group
        self.root = root
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    # fixed-size case: every rank contributes exactly `points_per_rank` points
    def gather_fixed(self, local_payload: torch.Tensor, stitched: Optional[torch.Tensor], async_op: bool = False):
        """local_payload: int16 [points*5]; stitched (root only): int16 [world*points*5]."""
        n = local_payload.numel()
        src = _as_bytes(local_payload)
        if self.rank == self.root:
            if stitched is None or stitched.numel() < n * self.world:
                raise ValueError("root needs a stitched buffer of world * local size")
            views = [_as_bytes(stitched[r * n:(r + 1) * n]) for r in range(self.world)]
            return dist.gather(src, views, dst=self.root, group=self.group, async_op=async_op)
        return dist.gather(src, None, dst=self.root, group=self.group, async_op=async_op)

    # variable-size case (after compaction)
    def gather_counts(self, local_points: int, device) -> List[int]:
        t = torch.tensor([int(local_points)], dtype=torch.int64, device=device)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return [int(x.item()) for x in out]

    def gather_variable(self, local_payload: torch.Tensor, local_points: int,
                        stitched: Optional[torch.Tensor]) -> List[int]:
        """Returns every rank's point count; on root, stitched[: sum*5] holds the concatenation."""
        counts = self.gather_counts(local_points, local_payload.device)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c * POINT_SHORTS)
        if self.rank == self.root:
            if stitched is None or stitched.numel() < offs[-1]:
                raise ValueError("stitched buffer too small")
            ops = []
            for r in range(self.world):
                if counts[r] == 0:
                    continue
                dst = stitched[offs[r]:offs[r + 1]]
                if r == self.root:
                    dst.copy_(local_payload[:counts[r] * POINT_SHORTS])
                else:
                    ops.append(dist.P2POp(dist.irecv, _as_bytes(dst), r, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        elif local_points > 0:
            ops = [dist.P2POp(dist.isend, _as_bytes(local_payload[:local_points * POINT_SHORTS]), self.root, self.group)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return counts
