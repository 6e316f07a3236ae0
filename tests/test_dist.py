"""CPU: the data-parallel sharding + box all-gather under gloo with world_size 2 (the N>1 path of
bench.py uses the same code over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from findtextcenternet_amd.dist import all_gather_boxes, shard_range


def test_shard_range_partitions_everything():
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                got += list(range(lo, hi))
            assert got == list(range(n))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, cap, Cf = 3, 16, 100
    g = torch.Generator().manual_seed(100 + rank)
    counts = torch.randint(0, cap, (B,), generator=g, dtype=torch.int32)
    boxes = torch.randn(B, cap, 9, generator=g)
    feats = torch.randn(B, cap, Cf, generator=g)
    out = all_gather_boxes(counts, boxes, feats)
    ok = out.counts.shape == (world * B,) and out.records.shape == (world * B, cap, 9 + Cf)
    for r in range(world):                                   # every rank sees every rank's records, in rank order
        gr = torch.Generator().manual_seed(100 + r)
        c = torch.randint(0, cap, (B,), generator=gr, dtype=torch.int32)
        bx = torch.randn(B, cap, 9, generator=gr)
        ft = torch.randn(B, cap, Cf, generator=gr)
        ok &= torch.equal(out.counts[r * B:(r + 1) * B], c)
        ok &= torch.equal(out.records[r * B:(r + 1) * B, :, :9], bx) and torch.equal(out.records[r * B:(r + 1) * B, :, 9:], ft)
    bt, ftile = out.tile(B)                                   # first tile of rank 1
    ok &= bt.shape[0] == int(out.counts[B])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_boxes_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_single_process_passthrough():
    counts = torch.tensor([2, 0], dtype=torch.int32)
    boxes, feats = torch.randn(2, 4, 9), torch.randn(2, 4, 100)
    out = all_gather_boxes(counts, boxes, feats)
    assert torch.equal(out.counts, counts) and torch.equal(out.records[..., :9], boxes)
