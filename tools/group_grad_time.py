"""Time the two gradient paths of grouping_operation (atomicAdd scatter vs fixed-order segmented sum) at the IA-SSD shapes.
usage: python tools/group_grad_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_batch_cuda as ext

dev = torch.device("cuda:0")
for (B, C, N, M, ns) in ((8, 24, 16384, 16384, 16), (8, 1, 16384, 4096, 32), (8, 64, 4096, 1024, 32), (8, 64, 4096, 1024, 16), (8, 128, 1024, 512, 32)):
    g = torch.Generator(device=dev).manual_seed(0)
    idx = torch.randint(0, N, (B, M, ns), generator=g, device=dev, dtype=torch.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]           # half-empty balls: repeats of the first hit
    go = torch.randn((B, C, M, ns), generator=g, device=dev)
    res = {}
    for label in ("atomic", "ordered"):
        out = torch.zeros((B, C, N), device=dev)
        def run():
            out.zero_()
            if label == "atomic":
                ext.group_points_grad_wrapper(B, C, N, M, ns, go, idx, out)
            else:
                ext.index_add_deterministic(go, idx, out)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        res[label] = (1e6 * (time.perf_counter() - t0) / 10, out.clone())
    err = (res["atomic"][1] - res["ordered"][1]).abs().max().item()
    print(f"B={B} C={C} N={N} M={M} ns={ns}: atomic {res['atomic'][0]:8.1f} us   ordered {res['ordered'][0]:8.1f} us   max diff {err:.1e}", flush=True)
