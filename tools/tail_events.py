"""HIP-event times of the three SA layers inside un-profiled sequential passes (what the kernel trace cannot show: the gaps).
GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import fused, pointnet2_modules as M, sa_stack, scenes
import spsnet_amd.pointnet2_batch_cuda as ext
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=0).to(dev)
xyz, feats = scenes.make_batch("kitti-lidar-v1", 8, 16384, seed0=0)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
marks = {}
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.setdefault(name, []).append(e)
orig_pub = ext.fps_publish
def pub(*a, **kw):
    ev("fps_start"); r = orig_pub(*a, **kw); ev("fps_end"); return r
ext.fps_publish = pub
for k in (1, 2):
    lay = layers[k]; fwd = lay.forward
    def wrapped(*a, _f=fwd, _k=k, **kw):
        ev(f"L{_k}_start"); r = _f(*a, **kw); ev(f"L{_k}_end"); return r
    lay.forward = wrapped
# (fps_start / fps_end are recorded on the PRODUCER's stream when the producer runs on a helper stream: the launch bracket)
for prec in ("fp16x2", "fp32"):
    fused.set_precision(prec)
    with torch.no_grad():
        for _ in range(3):
            sa_stack.run_sa_layers(layers, x, f)
        torch.cuda.synchronize(); marks.clear()
        for _ in range(20):
            ev("step_start"); sa_stack.run_sa_layers(layers, x, f); ev("step_end")
    torch.cuda.synchronize()
    def span(a, b):
        return float(np.median([s.elapsed_time(e) for s, e in zip(marks[a], marks[b])])) * 1e3
    print(prec, "step %.0f us | fps launch %.0f | step_start->L1_start %.0f | L1 %.0f | L1_end->L2_start %.0f | L2 %.0f | L2_end->step_end %.0f"
          % (span("step_start", "step_end"), span("fps_start", "fps_end"), span("step_start", "L1_start"), span("L1_start", "L1_end"),
             span("L1_end", "L2_start"), span("L2_start", "L2_end"), span("L2_end", "step_end")))
