"""Data-parallel sharding of tiles over the GPUs of one node and the RCCL all-gather of boxes.

The reference has no distributed code; tiles are independent units in eval
mode, so the batch is split into contiguous blocks, one process per GPU, weights replicated, and the
only exchange is the gather of the decoded peaks (never the 1 GB of heat-maps):

1. ``counts`` (one int32 per local tile, padded to the largest local batch) -- a few bytes per rank;
2. the record block ``[B_pad, n_max, 112]`` fp32 cut to ``n_max`` = the largest count on any rank (capped at the
   decode capacity): box (9 floats), 3 pad, the 100-d feature row -- exactly the rows ``ftc_decode`` wrote, no
   repacking (counts first, then pad to the global maximum).

``backend='nccl'`` is RCCL on ROCm; over xGMI a message of this size (about 0.7 MB per tile at 1600 peaks) is
latency-bound, so one ``all_gather_into_tensor`` per array is used.  The same code runs under ``gloo`` on CPU
tensors for the world_size-2 tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) block of items owned by `rank` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class GatheredBoxes:
    counts: torch.Tensor      # [n_tiles] int32 peaks found per tile, global tile order
    records: torch.Tensor     # [n_tiles, n_max, W] fp32 record rows (W = 112: box 0..8, feature row 12..111)
    feat0: int = 12
    message_bytes_per_rank: int = 0
    overflow: Optional[torch.Tensor] = None      # all_gather_boxes_static: device flag, a tile had more peaks than rows were sent

    def tile(self, i: int):
        n = min(int(self.counts[i]), self.records.shape[1])
        r = self.records[i, :n]
        return r[:, :9], r[:, self.feat0:]


def all_gather_boxes(counts: torch.Tensor, records: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                     feat0: int = 12) -> GatheredBoxes:
    """counts [B_local] int32, records [B_local, cap, W] fp32 (``Decoded.records``).  ``B_local`` may differ between
    ranks (``shard_range`` hands out uneven blocks when the tile count is not a multiple of the world size): every rank
    pads to the largest local batch with count-0 tiles and the padding is dropped again after the gather, so the result
    is in global tile order on every rank."""
    cap = records.shape[1]
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        n_max = min(cap, int(counts.max().item())) if counts.numel() else 0
        return GatheredBoxes(counts.clone(), records[:, :n_max].contiguous(), feat0, 0)
    world = dist.get_world_size(group)
    dev = counts.device
    B = counts.shape[0]
    # step 1: [B_local, max count] of every rank
    meta = torch.tensor([B, int(counts.max().item()) if B else 0], dtype=torch.int32, device=dev)
    metas = torch.empty((world * 2,), dtype=torch.int32, device=dev)          # flat output: what gloo and RCCL both accept
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas_h = metas.cpu().view(world, 2)
    b_pad = int(metas_h[:, 0].max())
    n_max = min(cap, int(metas_h[:, 1].max()))
    cnt = counts
    if B < b_pad:
        cnt = torch.cat([counts, torch.zeros(b_pad - B, dtype=counts.dtype, device=dev)])
    out_c = torch.empty((world * b_pad,), dtype=counts.dtype, device=dev)
    dist.all_gather_into_tensor(out_c, cnt.contiguous(), group=group)
    # step 2: only the rows any rank filled
    rec = records[:, :n_max]
    if B < b_pad:
        rec = torch.cat([rec, torch.zeros((b_pad - B,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=dev)])
    rec = rec.contiguous()                                   # a strided slice of the decode block: one copy of the live rows
    out_r = torch.empty((world * b_pad,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=dev)
    if rec.numel():
        dist.all_gather_into_tensor(out_r.view(-1), rec.view(-1), group=group)
    if (metas_h[:, 0] != b_pad).any():                       # drop the padding tiles of the short ranks
        keep = torch.cat([torch.arange(r * b_pad, r * b_pad + int(metas_h[r, 0])) for r in range(world)]).to(dev)
        out_c, out_r = out_c.index_select(0, keep), out_r.index_select(0, keep)
    return GatheredBoxes(out_c, out_r, feat0, rec.numel() * rec.element_size() + cnt.numel() * 4 + 8)


STATIC_GATHER_BYTES = 8 << 20      # per-rank message up to which the whole fixed-capacity record block is sent as it is


def static_gather_rows(n_tiles: int, world: int, cap: int, row_words: int, row_hint: Optional[int]) -> int:
    """Rows per tile that every rank sends in ``all_gather_boxes_static`` -- computed from RANK-INDEPENDENT quantities only (the global
    tile count, the world size, the decode capacity and the caller's hint), never from the local shard: ``shard_range`` hands out uneven
    shards, and a decision taken on the local block size can land two ranks on opposite sides of ``STATIC_GATHER_BYTES`` -- RCCL does not
    check message sizes, the collective would hang or corrupt memory.  The whole capacity when the LARGEST shard's block is small, else
    the hint (None: the whole capacity)."""
    b_pad = max(1, -(-int(n_tiles) // max(1, int(world))))
    if row_hint is None or b_pad * cap * row_words * 4 <= STATIC_GATHER_BYTES:
        return cap
    return max(1, min(cap, int(row_hint)))


def all_gather_boxes_static(counts: torch.Tensor, records: torch.Tensor, n_tiles: int, group: Optional[dist.ProcessGroup] = None,
                            feat0: int = 12, rows: Optional[int] = None, row_hint: Optional[int] = None) -> GatheredBoxes:
    """The steady-state form of ``all_gather_boxes``: NO host synchronisation, ONE collective, no copy of the record block.

    The counts-first protocol above needs two host round trips per step (``.item()`` for the row count, ``.cpu()`` for the other
    ranks' batch sizes), which serialise the host enqueue behind the GPU every step.  Here everything the host needs is static:
    ``n_tiles`` (the global tile count; ``shard_range`` gives every rank's share, so the padding to the largest shard is known
    without asking), and the row count -- the whole decode capacity when the block is small (8 tiles x 2048 rows x 448 B = 7.3 MB:
    latency-bound on xGMI either way), else ``row_hint`` (e.g. last step's GLOBAL maximum, rounded up; it must be the same number on
    every rank), with ``overflow`` (a device flag, identical on every rank) telling afterwards whether a tile had more peaks than were
    sent.  ``rows`` forces a row count (tests; the caller guarantees it is rank-independent).  The per-tile counts travel INSIDE the
    block, in the first padding word (column 9) of every tile's row 0 -- the box occupies columns 0..8, the feature row starts at
    ``feat0`` = 12.  NOTE: that word is written IN PLACE into the caller's ``records`` (also when there is only one rank)."""
    cap = records.shape[1]
    world_ = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    n_rows = (max(1, min(cap, int(rows))) if rows is not None
              else static_gather_rows(n_tiles, world_, cap, records.shape[2], row_hint))
    records[:, 0, 9] = counts.view(torch.float32)                  # int32 bit patterns in a padding word (in place: rows are the caller's scratch)
    block = records if n_rows == cap else records[:, :n_rows].contiguous()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        g = GatheredBoxes(counts, block, feat0, 0)
        g.overflow = (counts > n_rows).any()
        return g
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shares = [shard_range(n_tiles, r, world) for r in range(world)]
    b_pad = max(hi - lo for lo, hi in shares)
    B = counts.shape[0]
    if B != shares[rank][1] - shares[rank][0]:
        raise ValueError("all_gather_boxes_static: the local batch is not this rank's shard_range share of n_tiles")
    if B < b_pad:                                                   # short shard: count-0 tiles (zeros: column 9 of row 0 reads as count 0)
        block = torch.cat([block, torch.zeros((b_pad - B,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)])
    out = torch.empty((world * b_pad,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out.view(-1), block.view(-1), group=group)
    if any(hi - lo != b_pad for lo, hi in shares):                  # static index list: no device data needed
        keep = torch.tensor([r * b_pad + i for r, (lo, hi) in enumerate(shares) for i in range(hi - lo)], device=out.device)
        out = out.index_select(0, keep)
    cnt = out[:, 0, 9].contiguous().view(torch.int32)
    g = GatheredBoxes(cnt, out, feat0, block.numel() * block.element_size())
    g.overflow = (cnt > n_rows).any()
    return g


# ----------------------------------------------------------------------------------------------------------------------
# Train step (BASELINE configs[4]: "train1.py step ... 2xMI355X DDP"): data-parallel gradient all-reduce.  The reference has no
# distributed training code; this is the build's own: one process per GPU, per-rank BatchNorm statistics (the
# reference uses plain BatchNorm2d), gradients summed over ranks.  TrainStep keeps every gradient in ONE flat fp32 buffer in
# parameter order; the backward pass completes it from the END (decoder, heads) towards the stem, so the buffer is cut into
# buckets from the end and each bucket is all-reduced on a side stream as soon as the ops that write into it have run -- the
# collective of bucket k overlaps the backward kernels of bucket k+1.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring
# all-reduce of 1.05 GB of gradients is per-link bound (~2 * 1.05 GB * (N-1)/N / link rate), so FEW LARGE buckets (default
# 256 MB) rather than DDP's 25 MB ones: the per-collective latency is paid 4-5 times per step instead of 40.
# ----------------------------------------------------------------------------------------------------------------------
class BucketedAllReduce:
    """Sums a flat gradient buffer over the ranks of `group` in buckets cut from the END of the buffer.  Device-agnostic (RCCL on
    GPU tensors, gloo on CPU tensors in the tests).  Averaging is the caller's business (TrainStep scales the loss by 1/world)."""

    def __init__(self, flat: torch.Tensor, bucket_bytes: int = 256 << 20, group: Optional[dist.ProcessGroup] = None, align: int = 4):
        if flat.dim() != 1 or not flat.is_contiguous():
            raise ValueError("BucketedAllReduce needs a contiguous 1-d buffer")
        self.flat, self.group = flat, group
        n = flat.numel()
        per = max(align, (bucket_bytes // flat.element_size()) // align * align)
        self.ranges = []                       # [(lo, hi)] element ranges, LAST part of the buffer first
        hi = n
        while hi > 0:
            lo = max(0, hi - per)
            self.ranges.append((lo, hi))
            hi = lo
        self.handles = []

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def reduce_bucket(self, i: int, async_op: bool = True):
        lo, hi = self.ranges[i]
        if self.world == 1:
            return None
        h = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            self.handles.append(h)
        return h

    def wait(self) -> None:
        for h in self.handles:
            h.wait()
        self.handles = []

    def reduce_all(self) -> None:
        for i in range(len(self.ranges)):
            self.reduce_bucket(i, async_op=True)
        self.wait()

    def message_bytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()
