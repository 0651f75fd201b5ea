"""Host-side enqueue time of one pass of the SA stack vs its device time (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=0).to(dev)
xyz, feats = scenes.make_batch("kitti-lidar-v1", 8, 16384, seed0=0)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
with torch.no_grad():
    for _ in range(3):
        sa_stack.run_sa_layers(layers, x, f)
    torch.cuda.synchronize()
    for label, kw in (("streamed", {}), ("sequential", {"stream_first_layer": False})):
        host = []
        t0 = time.perf_counter()
        for _ in range(20):
            h0 = time.perf_counter()
            sa_stack.run_sa_layers(layers, x, f, **kw)
            host.append(time.perf_counter() - h0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{label}: host enqueue per pass median {sorted(host)[10]*1e3:.3f} ms (min {min(host)*1e3:.3f}); "
              f"20 passes enqueued in {(t1-t0)*1e3:.1f} ms, finished after {(t2-t0)*1e3:.1f} ms")
