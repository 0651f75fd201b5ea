"""Time the grid ball query (csrc/ball_query_grid.hip) against the scan at the DenseEdgeConv size.
usage: python tools/bqg_time.py [B] [N] [M]   (M defaults to N: self query)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import scenes, pointnet2_batch_cuda as ext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
M = int(sys.argv[3]) if len(sys.argv) > 3 else N
dev = torch.device("cuda:0")
xyz, _ = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
x = torch.from_numpy(xyz).to(dev)
q = x if M == N else x[:, torch.randperm(N, device=dev)[:M]].contiguous()
for r, ns in ((0.8, 16), (0.2, 16), (0.8, 32), (1.6, 32)):
    res = {}
    for label, gate in (("grid", (0, 0)), ("scan", None)):
        ext.BQ_GRID_MIN = gate
        idx = torch.zeros((B, M, ns), dtype=torch.int32, device=dev)
        for _ in range(5):
            ext.ball_query_wrapper(B, N, M, r, ns, q, x, idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ext.ball_query_wrapper(B, N, M, r, ns, q, x, idx)
        torch.cuda.synchronize()
        res[label] = (1e6 * (time.perf_counter() - t0) / 20, idx.clone())
    assert torch.equal(res["grid"][1], res["scan"][1])
    print(f"B={B} N={N} M={M} r={r} ns={ns}: grid {res['grid'][0]:8.1f} us   scan {res['scan'][0]:8.1f} us", flush=True)
