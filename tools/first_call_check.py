import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import fused, pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
def pollute():
    junk = [torch.full((1 << 22,), 7.5, device=dev) for _ in range(8)] + [torch.full((1 << 20,), 3, dtype=torch.int32, device=dev) for _ in range(8)]
    del junk
for label, B, N, npts, ns, half in (("fp32 16k", 2, 16384, None, None, False), ("half ns64", 2, 8192, [2048, 512, 128], [(64, 64)] * 3, True),
                                    ("fp32 ns64", 2, 8192, [2048, 512, 128], [(64, 64)] * 3, False)):
    for trial in range(3):
        cfg = sa_stack.scaled_config(npoints=npts, nsamples=ns)
        layers = sa_stack.build_sa_layers(M, cfg, seed=2 + trial).to(dev)
        ref_layers = copy.deepcopy(layers)
        xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=31)
        x = torch.from_numpy(xyz).to(dev)
        f = torch.from_numpy(feats.astype(np.float16) if half else feats).to(dev)
        pollute()
        torch.cuda.synchronize()
        with torch.no_grad():
            got = sa_stack.run_sa_layers(layers, x, f)          # FIRST call on these modules: streamed
            torch.cuda.synchronize()
            ref = sa_stack.run_sa_layers(ref_layers, x, f, overlap=False, stream_first_layer=False)
            torch.cuda.synchronize()
        msg = []
        for k in range(3):
            for name, a, b in zip(("xyz", "feat", "cls", "idx"), got[k], ref[k]):
                if a is not None and not torch.equal(a, b):
                    bad = (a != b); where = bad.nonzero()
                    msg.append(f"L{k}.{name}: {int(bad.sum())} differ, first {where[0].tolist()} last {where[-1].tolist()}")
        print(label, "trial", trial, msg or "identical")
