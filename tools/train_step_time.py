"""Time a TRAINING step (forward + backward, BatchNorm on batch statistics) of the IA-SSD SA layers 0-2 through the
op-by-op path (HIP sampling / query / group kernels + torch Conv/BN), and print the top kernels.
usage: python tools/train_step_time.py [B] [N] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=3).to(dev).train()
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
x = torch.from_numpy(xyz).to(dev)
f = torch.from_numpy(feats).to(dev)


def step():
    for p in layers.parameters():
        p.grad = None
    outs = sa_stack.run_sa_layers(layers, x, f)
    loss = sum(o[1].square().mean() for o in outs) + sum(o[2].square().mean() for o in outs if o[2] is not None)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
print(f"training step (SA L0-L2, fwd+bwd) {B}x{N}: {ms:.2f} ms ({B * N / ms / 1e3:.2f} M points/s)", flush=True)
