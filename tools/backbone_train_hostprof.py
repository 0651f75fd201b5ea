"""Where the HOST time of an IASSD_Backbone training step goes: forward / backward enqueue times on an idle GPU, a cProfile
of the forward (calling thread) and of the backward (autograd's device thread, via threading.setprofile)."""
import os, sys, time, threading, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import backbones as BB, scenes

B, N = 8, 16384
dev = torch.device("cuda:0")
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
bidx = np.repeat(np.arange(B, dtype=np.float32), N)[:, None]
points = torch.from_numpy(np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)).to(dev)
net = scenes.fill_parameters(BB.IASSD_Backbone(BB.IASSD_KITTI_CFG, num_class=3, input_channels=4), 5).to(dev).train()

def fwd():
    for p in net.parameters():
        p.grad = None
    out = net(dict(batch_size=B, points=points))
    loss = out["centers_features"].square().mean() + out["ctr_offsets"][:, 1:].square().mean()
    for t in out["sa_ins_preds"]:
        if isinstance(t, torch.Tensor):
            loss = loss + t[..., 1:].square().mean()
    return loss

for _ in range(3):
    fwd().backward()
tf = tb = 0.0
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); loss = fwd(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter()
    tf += t1 - t0; tb += t3 - t2
torch.cuda.synchronize()
print(f"host enqueue: forward {tf * 100:.2f} ms, backward {tb * 100:.2f} ms", flush=True)

pr = cProfile.Profile(); pr.enable()
losses = [fwd() for _ in range(5)]
pr.disable()
print("---- forward, calling thread (5 steps)")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)

prof_b = cProfile.Profile()
def hook(*a):
    threading.setprofile(None)
    prof_b.enable()
threading.setprofile(hook)          # threads started from here on (autograd's device thread is created lazily per process: may
for l in losses:                    # already exist -> then only the calling thread's share shows)
    l.backward()
threading.setprofile(None)
prof_b.disable()
print("---- backward (5 steps)")
try:
    pstats.Stats(prof_b).sort_stats("tottime").print_stats(22)
except Exception as e:
    print("no backward profile:", e)
