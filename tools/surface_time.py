"""Time FeatureExtraction (surface_feature.py:119-187) at the generator's size, fused vs op-by-op.
usage: python tools/surface_time.py [B] [N]"""
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
    net = SF.FeatureExtraction(dynamic_graph=(mode == "dynamic")).to(dev).eval()
    outs = {}
    for label, grad in (("fused", False), ("op_by_op", True)):
        ctx = torch.enable_grad() if grad else torch.no_grad()
        SF.FUSED_TRAINING = not grad          # grad mode with the fused training kernels off = the reference's op sequence
        with ctx:
            for _ in range(5):
                out = net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10 if not grad else 3
            for _ in range(reps):
                out = net(x)
            torch.cuda.synchronize()
        outs[label] = out.detach()
        print(f"{mode:8s} {label:9s} {1e3 * (time.perf_counter() - t0) / reps:8.3f} ms per forward ({B}x{N})", flush=True)
    scale = outs["op_by_op"].abs().max().item()
    diff = (outs["fused"] - outs["op_by_op"]).abs().amax(dim=-1) / scale
    print(f"{mode:8s} max |fused - op_by_op| / max |op_by_op| = {diff.max().item():.2e}; points off by more than 1e-4: "
          f"{(diff > 1e-4).float().mean().item():.2e} of {diff.numel()}", flush=True)
SF.FUSED_TRAINING = True
