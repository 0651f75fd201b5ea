import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=3).to(dev).train()
xyz, feats = scenes.make_batch("kitti-lidar-v1", 8, 16384, seed0=1)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
def step(prefetch=False):
    for p in layers.parameters(): p.grad = None
    outs = sa_stack.run_sa_layers(layers, x, f)
    loss = sum(o[1].square().mean() for o in outs) + sum(o[2].square().mean() for o in outs if o[2] is not None)
    if prefetch: sa_stack.prefetch_first_layer(layers, x)
    loss.backward()
for flag in (True, False):
    M.FUSED_MLP_TRAINING = flag
    for _ in range(3): step()
    host = 0.0
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); host += time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"fused={flag}: host enqueue {host / 10 * 1e3:.2f} ms per step (GPU idle at the start of each)")
import cProfile, pstats
M.FUSED_MLP_TRAINING = True
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
