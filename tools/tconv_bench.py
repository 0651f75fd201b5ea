"""Throughput of the fused train-mode convolution kernel (csrc/mlp_train.hip) at fixed bytes and varying row length: is the
~2.2 TB/s ceiling of every channel-major kernel (many rows x short contiguous runs) a property of the access pattern?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spsnet_amd.pointnet2_batch_cuda as ext
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for ci, co in ((32, 64), (64, 64), (128, 128), (256, 256)):
    for B, L in ((8, 131072), (64, 16384), (512, 2048)):
        x = torch.randn(B, ci, L, device=dev)
        y = torch.empty(B, co, L, device=dev)
        w = torch.randn(co, ci, device=dev) * 0.1
        wa = w.abs().amax().reshape(1)
        P = torch.zeros(ci, 8, device=dev); P[:, 2] = 1.0
        t_raw = timeit(lambda: ext.tconv(w, wa, ext.TIN_RAW, ext.TEPI_STATS, y, operand=x))
        t_bn = timeit(lambda: ext.tconv(w, wa, ext.TIN_BNRELU, ext.TEPI_STATS, y, operand=x, pin=P))
        z = torch.empty_like(x)
        t_copy = timeit(lambda: z.copy_(x))
        gb = (x.numel() + y.numel()) * 4 / 1e9
        print(f"ci={ci:3d} co={co:3d} B={B:3d} L={L:6d}: raw {gb / t_raw:6.0f} GB/s  bn+relu {gb / t_bn:6.0f} GB/s  "
              f"(torch copy {2 * x.numel() * 4 / 1e9 / t_copy:6.0f} GB/s)", flush=True)

print("weight gradient (twgrad): dY from (dA, y), the other operand through BatchNorm + ReLU")
for co, ci in ((32, 32), (64, 32), (128, 128), (256, 256)):
    for B, L in ((8, 131072), (8, 16384)):
        y = torch.randn(B, co, L, device=dev)
        dA = torch.randn(B, co, L, device=dev) * 1e-4
        x = torch.randn(B, ci, L, device=dev)
        pd = torch.zeros(co, 8, device=dev); pd[:, 1] = 1.0; pd[:, 2] = 1.0
        px = torch.zeros(ci, 8, device=dev); px[:, 2] = 1.0
        am = torch.full((1,), 4e-4, device=dev)
        t = timeit(lambda: ext.twgrad(y, pd, x, px, am, dA=dA))
        gb = (y.numel() * 2 + x.numel()) * 4 / 1e9
        print(f"co={co:3d} ci={ci:3d} B={B:3d} L={L:6d}: {gb / t:6.0f} GB/s  ({t * 1e6:6.1f} us)", flush=True)
