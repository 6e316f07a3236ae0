"""Data-parallel sharding of tiles over the GPUs of one node and the RCCL all-gather of boxes.

The reference has no distributed code; tiles are independent units in eval
mode, so the batch is split into contiguous blocks, one process per GPU, weights replicated, and the
only exchange is the gather of the decoded peaks (never the 1 GB of heat-maps): counts first, then
one fixed-capacity ``[B_local, cap, 9+C]`` fp32 record block per rank.  ``backend='nccl'`` is RCCL
on ROCm; over xGMI the message (a few MB) is latency-bound, so a single all_gather_into_tensor per
array is used.  The same code runs under ``gloo`` on CPU tensors for the world_size-2 tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) block of items owned by `rank` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class GatheredBoxes:
    counts: torch.Tensor      # [world * B_local] int32 peaks found per tile (global tile order)
    records: torch.Tensor     # [world * B_local, cap, 9 + C] fp32: box (9) ++ feature row (C)

    def tile(self, i: int):
        n = min(int(self.counts[i]), self.records.shape[1])
        r = self.records[i, :n]
        return r[:, :9], r[:, 9:]


def pack_records(boxes: torch.Tensor, feats: torch.Tensor) -> torch.Tensor:
    return torch.cat([boxes, feats], dim=2).contiguous()


def all_gather_boxes(counts: torch.Tensor, boxes: torch.Tensor, feats: torch.Tensor,
                     group: Optional[dist.ProcessGroup] = None) -> GatheredBoxes:
    """counts [B] int32, boxes [B,cap,9], feats [B,cap,C] (same B and cap on every rank)."""
    rec = pack_records(boxes, feats)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return GatheredBoxes(counts.clone(), rec)
    world = dist.get_world_size(group)
    out_c = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    out_r = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out_c, counts.contiguous(), group=group)
    dist.all_gather_into_tensor(out_r, rec, group=group)
    return GatheredBoxes(out_c, out_r)
