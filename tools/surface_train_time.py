"""Time FeatureExtraction forward + backward (training, op-by-op autograd path vs the fused Function when available).
usage: python tools/surface_train_time.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import scenes, surface_feature as SF

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda:0")
xyz, _ = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
x = torch.from_numpy(xyz).to(dev)
for mode in ("dynamic", "static"):
    torch.manual_seed(0)
    net = SF.FeatureExtraction(dynamic_graph=(mode == "dynamic")).to(dev).train()
    def step():
        for p in net.parameters():
            p.grad = None
        out = net(x)
        out.square().mean().backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(f"{mode:8s} forward+backward {1e3 * (time.perf_counter() - t0) / 3:8.2f} ms ({B}x{N})", flush=True)
