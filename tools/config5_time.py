"""BASELINE config 5 shape on ONE GPU: 1 scene x 180 000 points -> 16 384 / 4 096 / 1 024 centroids, nsample 64 (fp16
features), and the Waymo YAML shape (2 x 65 536 -> 16 384 / 4 096 / 2 048, nsample 16 & 32, fp32): per pass with the first
layer streamed behind the FPS producer, unstreamed, and the per-kernel events of one streamed pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
for label, B, N, npts, ns, half in (("config 5 (1 scene/GPU)", 1, 180000, [16384, 4096, 1024], [(64, 64)] * 3, True),
                                    ("Waymo YAML shape", 2, 65536, [16384, 4096, 2048], None, False)):
    cfg = sa_stack.scaled_config(npoints=npts, nsamples=ns)
    layers = sa_stack.build_sa_layers(M, cfg, seed=0).to(dev)
    xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
    if half:
        f = f.half()
    for streamed in (True, False):
        with torch.no_grad():
            for _ in range(2):
                sa_stack.run_sa_layers(layers, x, f, stream_first_layer=streamed)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                sa_stack.run_sa_layers(layers, x, f, stream_first_layer=streamed)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{label}: B={B} N={N} -> {npts} streamed={streamed}: {dt*1e3:.2f} ms/pass = {B*N/dt/1e6:.2f} M points/s", flush=True)
