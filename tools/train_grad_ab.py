import sys, os, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes
dev = torch.device("cuda:0")
base = sa_stack.build_sa_layers(M, sa_stack.scaled_config(npoints=[1024, 256, 128]), seed=9).to(dev).train()
xyz, feats = scenes.make_batch("kitti-lidar-v1", 2, 8192, seed0=77, dup_fraction=0.01)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feats).to(dev)
def run(streamed, fusedmlp):
    layers = copy.deepcopy(base)
    sa_stack.STREAM_TRAINING_QUERIES = streamed
    M.FUSED_MLP_TRAINING = fusedmlp
    outs = sa_stack.run_sa_layers(layers, x, f)
    loss = sum(o[1].square().mean() for o in outs) + sum(o[2].square().mean() for o in outs if o[2] is not None)
    loss.backward()
    torch.cuda.synchronize()
    return layers
def cmp(a, b, what):
    worst = {}
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None: continue
        e = float((pa.grad - pb.grad).abs().max() / max(1e-30, float(pb.grad.abs().max())))
        key = n.split(".")[0] + "." + n.split(".")[1]
        if e > worst.get(key, (-1, ""))[0]: worst[key] = (e, n)
    print(what, " ".join(f"{k}:{v[0]:.1e}({v[1].split('.', 2)[2]})" for k, v in sorted(worst.items())), flush=True)
for fm in (False, True):
    a, b = run(False, fm), run(False, fm)
    cmp(a, b, f"fused_mlp={fm}: two identical runs:")
