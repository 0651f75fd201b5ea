"""Time a training step (forward + backward) of IASSD_Backbone / PAGNet_Backbone at the KITTI configuration.
usage: python tools/backbone_train_time.py [B] [N] [reps] [IASSD|PAGNet] [prefetch]
  prefetch: net.prefetch_sampling(next batch) before every backward (the next batch's FPS + ball queries beside the backward)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import backbones as BB, scenes, surface_feature as SF

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
only = sys.argv[4] if len(sys.argv) > 4 else ""
PREFETCH = len(sys.argv) > 5 and sys.argv[5] == "prefetch"
dev = torch.device("cuda:0")
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
bidx = np.repeat(np.arange(B, dtype=np.float32), N)[:, None]
points = torch.from_numpy(np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)).to(dev)
stds = torch.from_numpy(np.random.default_rng(0).uniform(0, 40, (B, N)).astype(np.float32)).to(dev)
for tag, cls, cfg in (("IASSD_Backbone", BB.IASSD_Backbone, BB.IASSD_KITTI_CFG), ("PAGNet_Backbone", BB.PAGNet_Backbone, BB.SPSNET_KITTI_CFG),
                      ("PointNet2MSG", BB.PointNet2MSG, BB.POINTRCNN_KITTI_CFG)):
    if only and not tag.startswith(only):
        continue
    if tag == "PointNet2MSG":       # (PointRCNN's backbone: per-point features out, no votes / class scores)
        net = scenes.fill_parameters(cls(cfg, input_channels=4), 5).to(dev).train()
        def step():
            for p in net.parameters():
                p.grad = None
            net(dict(batch_size=B, points=points))["point_features"].square().mean().backward()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        print(f"{tag:16s} training step {B}x{N}: {1e3 * (time.perf_counter() - t0) / reps:7.2f} ms", flush=True)
        continue
    for fused_fe in ((True, False) if tag.startswith("PAG") and not only else (True,)):
        SF.FUSED_TRAINING = fused_fe
        net = scenes.fill_parameters(cls(cfg, num_class=3, input_channels=4), 5).to(dev).train()
        def step():
            for p in net.parameters():
                p.grad = None
            d = dict(batch_size=B, points=points)
            if tag.startswith("PAG"):
                d["stds"] = stds
            out = net(d)
            loss = out["centers_features"].square().mean() + out["ctr_offsets"][:, 1:].square().mean()
            for t in out["sa_ins_preds"]:
                if isinstance(t, torch.Tensor):
                    loss = loss + t[..., 1:].square().mean()
            if PREFETCH:
                net.prefetch_sampling(d)
            loss.backward()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        host = 0.0   # host side alone: the step's Python + autograd-engine time on an idle GPU
        for _ in range(reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            host += time.perf_counter() - t1
        torch.cuda.synchronize()
        note = "" if tag.startswith("IASSD") else (" (fused surface-feature training kernels)" if fused_fe else " (surface features op by op)")
        print(f"{tag:16s} training step {B}x{N}: {ms:7.2f} ms, host enqueue {1e3 * host / reps:.2f} ms{note}"
              + (" [next batch's sampling prefetched]" if PREFETCH else ""), flush=True)
SF.FUSED_TRAINING = True
