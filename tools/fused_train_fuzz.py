"""Random-shape sweep of the fused train-mode grouped MLP (csrc/mlp_train.hip) against float64 torch: depth 1-3, widths 1-256,
every nsample, column counts that are any multiple of 64, batch 1-3, gradients of random magnitude.  Prints the worst relative
error per quantity; exits non-zero on a miss (> 5e-5 of the quantity's own largest magnitude, near-tie arg-max flips aside)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import fused, pointnet2_modules as PM
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst, bad = {}, 0
for case in range(cases):
    ns = int(rng.choice([4, 8, 16, 32, 64]))
    B = int(rng.integers(1, 4))
    M = int(rng.integers(1, 40)) * 64 // ns if ns <= 64 else 1
    M = max(M, 64 // ns)
    while (M * ns) % 64:
        M += 1
    depth = int(rng.integers(1, 4))
    widths = [int(rng.integers(1, 257)) for _ in range(depth + 1)]
    gscale = float(10.0 ** rng.uniform(-9, 3))
    torch.manual_seed(case)
    mlp = PM._conv_bn_relu_stack(list(widths), torch.nn.Conv2d, torch.nn.BatchNorm2d)
    for mod in mlp:
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.3, 1.5)
            mod.bias.data.normal_(0, 0.3)
    ref = copy.deepcopy(mlp).double().train()
    mlp = mlp.to(dev).train()
    x0 = torch.randn(B, widths[0], M, ns) * float(rng.uniform(0.1, 5)) + float(rng.uniform(-1, 1))
    wout = torch.randn(B, widths[-1], M) * gscale
    xr = x0.double().requires_grad_(True)
    yr, cond = xr, 0.0
    for mod in ref:      # conditioning of the BatchNorms: |mean| / std of the pre-activation (the kernels' error is relative to |y|)
        yr = mod(yr)
        if isinstance(mod, torch.nn.Conv2d):
            yd = yr.detach()
            cond = max(cond, float((yd.mean(dim=(0, 2, 3)).abs() / yd.std(dim=(0, 2, 3)).clamp_min(1e-12)).max()))
    out_r = yr.max(dim=3)[0]
    (out_r * wout.double()).sum().backward()
    # near-ties of the arg-max (top two within 1e-5 relative) make the gradient routing ill-defined: skip such cases
    top2 = yr.detach().topk(2, dim=3)[0] if ns > 1 else None
    # (the kernels' 22-bit arithmetic is accurate relative to the LARGEST activation, so the gap is measured against that)
    tie = bool((((top2[..., 0] - top2[..., 1]) <= 2e-5 * (1.0 + cond) * float(yr.detach().abs().max())) & (top2[..., 0] > 0)).any())
    xg = x0.to(dev).requires_grad_(True)
    got = PM._fused_mlp_pool_train(mlp, xg, 'max_pool')
    assert got is not None, (B, M, ns, widths)
    (got * wout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert not fused.check_overflow(), (B, M, ns, widths)
    items = [("out", got, out_r)] + ([] if tie else [("dx", xg.grad, xr.grad)] + [("d" + n.split(".")[-1], p.grad, q.grad) for (n, p), (_, q) in
                                                                                   zip(mlp.named_parameters(), ref.named_parameters())])
    for name, a, b in items:
        err = float((a.detach().cpu().double() - b.detach()).abs().max() / max(1e-300, float(b.abs().max())))
        worst[name] = max(worst.get(name, 0.0), err)
        if err > 5e-5:
            bad += 1
            print("MISS", name, f"{err:.2e}", dict(B=B, M=M, ns=ns, widths=widths, gscale=gscale, tie=tie, cond=round(cond, 1)), flush=True)
print("cases", cases, "worst relative errors:", {k: f"{v:.1e}" for k, v in worst.items()}, "misses", bad, flush=True)
sys.exit(1 if bad else 0)
