"""Per-step HIP-event times of 60 sequential passes (distribution: mean / median / min / max).  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from spsnet_amd import fused, pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=0).to(dev)
xyz, feats = scenes.make_batch("kitti-lidar-v1", 8, 16384, seed0=0)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
for prec in ("fp16x2", "fp32"):
    fused.set_precision(prec)
    with torch.no_grad():
        for _ in range(5): sa_stack.run_sa_layers(layers, x, f)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
        marks[0].record()
        for i in range(60):
            sa_stack.run_sa_layers(layers, x, f); marks[i + 1].record()
        torch.cuda.synchronize()
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(60)]
    print(prec, "mean %.4f median %.4f min %.4f max %.4f" % (np.mean(ms), np.median(ms), min(ms), max(ms)))
    print("  ", " ".join("%.3f" % v for v in ms))
