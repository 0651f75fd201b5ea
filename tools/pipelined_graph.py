"""Several complete SA-stack passes IN FLIGHT, each a warm-captured HIP graph replayed on a stream of its own
(spsnet_amd.graphs): layer-0 FPS keeps 8 of 256 compute units busy for ~78 % of a pass, so independent batches overlap -- and
with one host call per pass (0.3 ms instead of 1.7 ms of Python enqueue) the host no longer bounds how many.
usage: python tools/pipelined_graph.py [in-flight list, e.g. 1,2,3,4,6] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import graphs, pointnet2_modules as M, sa_stack, scenes

flights = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,6").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B, N = 8, 16384
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=0).to(dev)
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=0)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
with torch.no_grad():
    ref = sa_stack.run_sa_layers(layers, x, f, overlap=False, stream_first_layer=False)
    torch.cuda.synchronize()
    for k in flights:
        slots = [graphs.graphed_sa_stack(layers, x, f) for _ in range(k)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(dev))
        for i in range(2 * k):
            with torch.cuda.stream(streams[i % k]):
                slots[i % k].graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % k]):
                slots[i % k].graph.replay()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        same = all((p is None and q is None) or torch.equal(p, q) for g in slots for la, lb in zip(g.static_out, ref) for p, q in zip(la, lb))
        print(f"{k} graph(s) in flight: {1e3 * el / steps:.3f} ms per pass = {B * N * steps / el / 1e6:.1f} M points/s "
              f"(host {1e3 * t_host / steps:.3f} ms per pass); every slot's outputs identical to a sequential eager pass: {same}; "
              f"timeouts {sa_stack.check_timeouts()}", flush=True)
        del slots
