"""Experiment: capture one pass of the SA stack into a HIP graph and replay it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=0).to(dev)
xyz, feats = scenes.make_batch("kitti-lidar-v1", 8, 16384, seed0=0)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
kw = {"stream_first_layer": sys.argv[1] != "seq"} if len(sys.argv) > 1 else {}
with torch.no_grad():
    for _ in range(3):
        ref = sa_stack.run_sa_layers(layers, x, f, **kw)
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f"eager : {timeit(lambda: sa_stack.run_sa_layers(layers, x, f, **kw)):.3f} ms/pass")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        sa_stack.run_sa_layers(layers, x, f, **kw)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = sa_stack.run_sa_layers(layers, x, f, **kw)
    torch.cuda.synchronize()
    print(f"graph : {timeit(g.replay):.3f} ms/pass")
    g.replay(); torch.cuda.synchronize()
    ok = all((a is None and b is None) or torch.equal(a, b) for la, lb in zip(out, ref) for a, b in zip(la, lb))
    print("graph outputs identical to eager:", ok, "timeouts:", sa_stack.check_timeouts())
