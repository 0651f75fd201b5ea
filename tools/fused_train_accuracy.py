"""Accuracy of the train-mode grouped MLP against a float64 CPU reference: the fused kernels (csrc/mlp_train.hip, split-fp16)
beside the op-by-op fp32 kernels; per quantity the error relative to that quantity's own largest magnitude."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import fused, pointnet2_modules as PM
dev = torch.device("cuda:0")
SHAPES = [(2, 128, 32, [131, 128, 256, 256]), (2, 1024, 32, [4, 32, 32, 64]), (2, 256, 16, [67, 64, 64, 128])]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")[:3]) + ([int(v) for v in a.split(",")[3:]],) for a in sys.argv[1:]]
for (B, M, ns, chain) in SHAPES:
    torch.manual_seed(1)
    mlp = PM._conv_bn_relu_stack(list(chain), torch.nn.Conv2d, torch.nn.BatchNorm2d)
    x0 = torch.randn(B, chain[0], M, ns) * 0.7 + 0.3
    wout = torch.randn(B, chain[-1], M)
    ref = copy.deepcopy(mlp).double().train()
    xr = x0.double().requires_grad_(True)
    out_ref = ref(xr).max(dim=3)[0]
    (out_ref * wout.double()).sum().backward()
    want = [out_ref.detach(), xr.grad] + [p.grad for p in ref.parameters()]
    names = ["out", "dx"] + ["d" + n for n, _ in ref.named_parameters()]
    for fused_on in (True, False):
        PM.FUSED_MLP_TRAINING = fused_on
        m2 = copy.deepcopy(mlp).to(dev).train()
        x = x0.to(dev).requires_grad_(True)
        out = PM._fused_mlp_pool_train(m2, x, 'max_pool')
        if out is None:
            out = PM._pool_over_samples(PM._shared_mlp(m2, x), 'max_pool')
        (out * wout.to(dev)).sum().backward()
        torch.cuda.synchronize()
        got = [out.detach(), x.grad] + [p.grad for p in m2.parameters()]
        errs = [(float((g.cpu().double() - w).abs().max() / w.abs().max()), n) for g, w, n in zip(got, want, names)]
        errs.sort(reverse=True)
        print((B, M, ns), chain, "fused   " if fused_on else "op-by-op", "overflow" if fused.check_overflow() else "", " ".join(f"{n}:{e:.1e}" for e, n in errs[:5]), flush=True)
