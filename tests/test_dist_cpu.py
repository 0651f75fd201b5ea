"""world_size-2 gloo test of the scene-sharding host logic (CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_covers_batch():
    from spsnet_amd.dist import shard_range
    for total in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spsnet_amd.dist import all_gather_sampled_idx, shard_range
        total = 8
        full = [torch.arange(total * m, dtype=torch.int32).reshape(total, m) * (l + 1) for l, m in enumerate((16, 8, 4))]
        b, e = shard_range(total, world, rank)
        got = all_gather_sampled_idx([t[b:e].contiguous() for t in full])
        ok = all(torch.equal(g, t) for g, t in zip(got, full)) and all(g.dtype == torch.int32 for g in got)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_all_gather_sampled_idx_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


def test_all_gather_single_process_is_identity():
    from spsnet_amd.dist import all_gather_sampled_idx
    a = [torch.arange(6, dtype=torch.int32).reshape(2, 3)]
    assert torch.equal(all_gather_sampled_idx(a)[0], a[0])
