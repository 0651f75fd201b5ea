"""Two ranks sharing one GPU (gloo collectives): the fused train-mode grouped MLP under nn.SyncBatchNorm with the batch split
over the ranks must reproduce ONE process running the whole batch through plain BatchNorm -- the definition of synchronised
batch statistics.  (RCCL refuses two ranks on one device; the collectives here are gloo's, the kernels the same.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _make(seed=21):
    from spsnet_amd import pointnet2_modules as PM
    torch.manual_seed(seed)
    mlp = PM._conv_bn_relu_stack([10, 32, 64, 48], torch.nn.Conv2d, torch.nn.BatchNorm2d)
    x = torch.randn(4, 10, 64, 16)
    wout = torch.randn(4, 48, 64) * 1e-2
    return mlp, x, wout


def _worker(rank, world, port, q, bounds):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spsnet_amd import pointnet2_modules as PM
        dev = torch.device("cuda:0")
        mlp, x, wout = _make()
        mlp = torch.nn.SyncBatchNorm.convert_sync_batchnorm(mlp).to(dev).train()
        lo, hi = bounds[rank], bounds[rank + 1]
        xs = x[lo:hi].to(dev).requires_grad_(True)
        PM.FUSED_MLP_TRAINING = True
        out = PM._fused_mlp_pool_train(mlp, xs, 'max_pool')
        assert out is not None
        (out * wout[lo:hi].to(dev)).sum().backward()
        torch.cuda.synchronize()
        grads = [p.grad.detach().cpu().double() for p in mlp.parameters()]
        for g in grads:                                   # what DDP does with parameter gradients (sum here, not mean)
            dist.all_reduce(g)
        # numpy, not tensors: a tensor travels as a file descriptor its (by then exited) producer has to hand over
        q.put((rank, out.detach().cpu().numpy(), xs.grad.detach().cpu().numpy(), [g.float().numpy() for g in grads],
               [b.detach().cpu().float().numpy() for b in mlp.buffers()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("bounds", [(0, 2, 4), (0, 3, 4)], ids=["equal-shards", "unequal-shards"])
def test_two_ranks_sync_batchnorm_equal_one_process_plain_batchnorm(bounds):
    """bounds = the scenes each rank holds: (0, 3, 4) gives the ranks DIFFERENT element counts -- the global count rides in
    the statistics' all-reduce (torch's SyncBatchNorm gathers per-rank counts the same way)."""
    from spsnet_amd import pointnet2_modules as PM
    dev = torch.device("cuda:0")
    mlp, x, wout = _make()
    mlp = mlp.to(dev).train()
    xg = x.to(dev).requires_grad_(True)
    old = PM.FUSED_MLP_TRAINING
    PM.FUSED_MLP_TRAINING = True
    try:
        want = PM._fused_mlp_pool_train(mlp, xg, 'max_pool')
        (want * wout.to(dev)).sum().backward()
    finally:
        PM.FUSED_MLP_TRAINING = old
    torch.cuda.synchronize()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, bounds)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        item = q.get(timeout=300)
        results[item[0]] = item[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0

    def close(a, b, what, tol=2e-5):
        err = float((a.double() - b.double()).abs().max())
        assert err <= tol * max(1e-30, float(b.abs().max())), (what, err, float(b.abs().max()))

    for r in range(2):
        lo, hi = bounds[r], bounds[r + 1]
        out, dx, grads, bufs = results[r]
        out, dx = torch.from_numpy(out), torch.from_numpy(dx)
        grads, bufs = [torch.from_numpy(g) for g in grads], [torch.from_numpy(b) for b in bufs]
        close(out, want[lo:hi].detach().cpu(), f"rank {r} output")
        close(dx, xg.grad[lo:hi].cpu(), f"rank {r} input gradient")
        for (name, p), g in zip(mlp.named_parameters(), grads):
            close(g, p.grad.cpu(), f"rank {r} summed gradient of {name}")
        for (name, b), bb in zip(mlp.named_buffers(), bufs):
            close(bb, b.detach().cpu().float(), f"rank {r} buffer {name}")
