"""Functional smoke of the other BASELINE configs on one GPU (not the bench): config 4 (sss_aware sampler) at full
size against the CPU oracle stack, and a Waymo-shaped single scene (65 536 points, nsample 64) for shapes/finite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
from oracle import cpu_stack
dev = torch.device("cuda:0")
# config 4: stability top-k at layer 2
cfg = sa_stack.scaled_config(sample_methods=['D-FPS', 'D-FPS', 'sss_aware'])
layers = sa_stack.build_sa_layers(M, cfg, seed=0)
xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 16384, seed0=5)
stds = np.random.default_rng(1).uniform(0, 40, (2, 16384)).astype(np.float32)
want = cpu_stack.sa_stack_cpu(cpu_stack.cpu_copy(layers), xyz, feats, stds)
layers = layers.to(dev)
with torch.no_grad():
    got = sa_stack.run_sa_layers(layers, torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(stds).to(dev))
for k in range(3):
    same = np.array_equal(got[k][3].cpu().numpy(), want[k][3])
    err = float(np.abs(got[k][1].cpu().numpy() - want[k][1]).max()) if same else float("nan")
    print(f"config4 layer {k}: idx {'exact' if same else 'differs (near-tie scores)'}  max|dfeat| {err:.2e}")
# Waymo-shaped: 65536 -> 16384 -> 4096 -> 1024, nsample 64
cfg5 = sa_stack.scaled_config(npoints=[16384, 4096, 1024], nsamples=[[64, 64]] * 3)
l5 = sa_stack.build_sa_layers(M, cfg5, seed=1).to(dev)
x5, f5 = scenes.make_batch("uniform-v1", 1, 65536, seed0=9)
torch.cuda.synchronize(); t0 = time.time()
with torch.no_grad():
    o5 = sa_stack.run_sa_layers(l5, torch.from_numpy(x5).to(dev), torch.from_numpy(f5).to(dev))
torch.cuda.synchronize()
print("waymo-shaped:", [tuple(o[1].shape) for o in o5], "finite", all(bool(torch.isfinite(o[1]).all()) for o in o5), f"{time.time()-t0:.2f}s")
