"""Time the stability generator's inference path (spsnet_amd/stability_generator.py) at its shipped size.
usage: python tools/generator_time.py [B] [N] [reps]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import pointnet2_batch_cuda as ext, scenes, stability_generator as SG

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
cfg = copy.deepcopy(SG.SF_UNC_CFG)
cfg['SA_CONFIG']['NPOINT_LIST'] = [[N]]
net = scenes.fill_parameters(SG.Generate_center(cfg), 5).to(dev).eval()
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
bidx = np.repeat(np.arange(B, dtype=np.float32), N)[:, None]
points = torch.from_numpy(np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)).to(dev)
for label, gate in (("grid ball query", (4096, 4096)), ("scan ball query", None)):
    ext.BQ_GRID_MIN = gate
    with torch.no_grad():
        for _ in range(5):
            net(dict(batch_size=B, points=points))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = net(dict(batch_size=B, points=points))
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    print(f"Generate_center (eval) {B}x{N}, {label}: {ms:7.3f} ms per forward ({B * N / ms / 1e3:.1f} M points/s)", flush=True)
