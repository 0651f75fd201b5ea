"""Error of the fp16-feature path (features stored as halves, grouped MLP on fp16 MFMA) against the fp32 CPU oracle stack:
prints per layer the max / mean absolute error relative to the layer's largest feature.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import cpu_stack
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes

dev = torch.device("cuda:0")
cases = [("small", 2, 4096, [1024, 256, 128], None), ("ns64", 2, 8192, [2048, 512, 128], [(64, 64)] * 3)]
if len(sys.argv) > 1 and sys.argv[1] == "full":
    cases.append(("config5", 1, 180000, [16384, 4096, 1024], [(64, 64)] * 3))
for label, B, N, npts, ns in cases:
    cfg = sa_stack.scaled_config(npoints=npts, nsamples=ns)
    layers = sa_stack.build_sa_layers(M, cfg, seed=2)
    xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=31)
    feats_h = feats.astype(np.float16)
    t0 = time.time()
    want = cpu_stack.sa_stack_cpu(cpu_stack.cpu_copy(layers), xyz, feats_h.astype(np.float32))
    t1 = time.time()
    layers = layers.to(dev)
    with torch.no_grad():
        got = sa_stack.run_sa_layers(layers, torch.from_numpy(xyz).to(dev), torch.from_numpy(feats_h).to(dev))
    torch.cuda.synchronize()
    print(f"{label}: cpu oracle {t1 - t0:.1f} s")
    for k, ((gx, gf, gc, gi), (wx, wf, wc, wi)) in enumerate(zip(got, want)):
        same = np.array_equal(gi.cpu().numpy(), wi)
        assert gf.dtype == torch.float16
        if same:
            err = np.abs(gf.float().cpu().numpy() - wf)
            scale = np.abs(wf).max()
            print(f"  layer {k}: idx exact; feature err max {err.max() / scale:.2e} mean {err.mean() / scale:.2e} (scale {scale:.3g})"
                  + (f"; cls err max {np.abs(gc.cpu().numpy() - wc).max():.2e}" if wc is not None else ""))
        else:
            common, gp, wp = np.intersect1d(gi.cpu().numpy()[0], wi[0], return_indices=True)
            err = np.abs(gf.float().cpu().numpy()[0][:, gp] - wf[0][:, wp])
            print(f"  layer {k}: {len(common) / wi.shape[1]:.3f} of scene 0's picks shared; matched feature err max "
                  f"{err.max() / np.abs(wf).max():.2e} mean {err.mean() / np.abs(wf).max():.2e}")
