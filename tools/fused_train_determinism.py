"""Run-to-run reproducibility of the fused train-mode grouped MLP (csrc/mlp_train.hip) and its sensitivity to noise in the
incoming gradient (what the LDS atomics of the grouping backward upstream produce), against the op-by-op kernels."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_modules as PM
dev = torch.device("cuda:0")
SHAPES = [(2, 1024, 16, [4, 16, 16, 32]), (2, 256, 32, [131, 128, 256, 256]), (8, 4096, 32, [4, 32, 32, 64])]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")[:3]) + ([int(v) for v in a.split(",")[3:]],) for a in sys.argv[1:]]
for (B, M, ns, chain) in SHAPES:
    torch.manual_seed(0)
    mlp = PM._conv_bn_relu_stack(list(chain), torch.nn.Conv2d, torch.nn.BatchNorm2d).to(dev).train()
    x0 = torch.randn(B, chain[0], M, ns, device=dev).abs() * 3 + 0.5       # (post-ReLU-like inputs: large positive mean)
    x0[:, :, :, ns // 2:] = x0[:, :, :, :1]                                  # (padded balls: repeated columns)
    wout = torch.randn(B, chain[-1], M, device=dev)
    noise = 1 + 1e-7 * torch.randn_like(wout)
    names = ["out", "dx"] + ["d" + n for n, _ in mlp.named_parameters()]
    for fused_on in (True, False):
        PM.FUSED_MLP_TRAINING = fused_on
        res = []
        for rep in range(3):
            m2 = copy.deepcopy(mlp)
            x = x0.clone().requires_grad_(True)
            out = PM._fused_mlp_pool_train(m2, x, 'max_pool')
            if out is None:
                out = PM._pool_over_samples(PM._shared_mlp(m2, x), 'max_pool')
            (out * (wout * noise if rep == 2 else wout)).sum().backward()
            torch.cuda.synchronize()
            res.append([out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m2.parameters()])
        bad = [n for n, a, b in zip(names, res[0], res[1]) if not torch.equal(a, b)]
        sens = max((float((a - b).abs().max() / a.abs().max()), n) for a, b, n in zip(res[0][1:], res[2][1:], names[1:]))
        print(chain, "fused" if fused_on else "op-by-op", "| differs between identical runs:", bad,
              "| largest relative change under 1e-7 gradient noise: %.2e (%s)" % sens, flush=True)
