"""Capture a pass into a HIP graph (torch.cuda.graph) and replay it -- on the captured input and on another batch copied into the
static input tensors -- against eager sequential passes: bit-identical outputs required.  By default the captured pass is the
FIRST pass of a fresh process (nothing of the library has run: the capture then also records the one-time weight packing);
GRAPH_TRY_WARM=1 runs three eager passes first (weights packed, helper streams placed and probed) and captures a WARM pass --
the honest replay-vs-eager comparison.  GRAPH_TRY_HOST=1 also prints the host time of a replay call and of an eager enqueue.
usage: python tools/graph_try.py [streamed|seq] [fp32|fp16x2]      (GPU box only; tests/test_parity_gpu.py runs it as a subprocess)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spsnet_amd
from spsnet_amd import fused, pointnet2_modules as M, sa_stack, scenes

mode = sys.argv[1] if len(sys.argv) > 1 else "streamed"
if len(sys.argv) > 2:
    fused.set_precision(sys.argv[2])
B = int(os.environ.get("GRAPH_TRY_B", "8"))
dev = torch.device("cuda:0")
kw = {"stream_first_layer": mode != "seq"}
spsnet_amd.init(dev)        # the library's ONE allocating / synchronising call: before the capture, never inside it
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=0).to(dev)
batches = [scenes.make_batch("kitti-lidar-v1", B, 16384, seed0=s) for s in (0, 100)]
x = torch.from_numpy(batches[0][0]).to(dev)
f = torch.from_numpy(batches[0][1]).to(dev)


def same(a, b):
    return all((p is None and q is None) or torch.equal(p, q) for la, lb in zip(a, b) for p, q in zip(la, lb))


WARM = os.environ.get("GRAPH_TRY_WARM", "0") == "1"
with torch.no_grad():
    if WARM:
        for _ in range(3):
            sa_stack.run_sa_layers(layers, x, f, **kw)
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):                      # (cold: nothing of this library has run in this process yet)
        out = sa_stack.run_sa_layers(layers, x, f, **kw)
    torch.cuda.synchronize()
    ok = True
    for k, (xa, fa) in enumerate(batches):
        x.copy_(torch.from_numpy(xa)); f.copy_(torch.from_numpy(fa))
        g.replay()
        torch.cuda.synchronize()
        got = [tuple(None if t is None else t.clone() for t in la) for la in out]
        ref = sa_stack.run_sa_layers(layers, x.clone(), f.clone(), overlap=False, stream_first_layer=False)
        torch.cuda.synchronize()
        ok_k = same(got, ref)
        print(f"replay on batch {k}: identical to an eager sequential pass: {ok_k}; timeouts {sa_stack.check_timeouts()}")
        ok &= ok_k

    def timeit(fn, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    for _ in range(3):
        sa_stack.run_sa_layers(layers, x, f, **kw)
    print(f"[{mode}, {fused.PRECISION}, {'warm' if WARM else 'first-pass'} capture, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}] "
          f"eager {timeit(lambda: sa_stack.run_sa_layers(layers, x, f, **kw)):.3f} ms/pass, graph replay {timeit(g.replay):.3f} ms/pass")
    if os.environ.get("GRAPH_TRY_HOST", "0") == "1":
        def host(fn, n=20):
            ts = []
            for _ in range(n):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
                torch.cuda.synchronize()
            return sorted(ts)[n // 2] * 1e3
        print(f"host time per pass (median of 20, device idle at each start): eager enqueue "
              f"{host(lambda: sa_stack.run_sa_layers(layers, x, f, **kw)):.3f} ms, graph replay call {host(g.replay):.3f} ms")
print("GRAPH_TRY_OK" if ok else "GRAPH_TRY_MISMATCH")
sys.exit(0 if ok else 1)
