"""Step-by-step check of the fused train-mode backward against float64 autograd intermediates (debugging aid)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import fused, pointnet2_modules as PM
import spsnet_amd.pointnet2_batch_cuda as ext
dev = torch.device("cuda:0")
B, M, ns = (int(v) for v in sys.argv[1:4])
chain = [int(v) for v in sys.argv[4:]]
torch.manual_seed(1)
mlp = PM._conv_bn_relu_stack(list(chain), torch.nn.Conv2d, torch.nn.BatchNorm2d)
x0 = torch.randn(B, chain[0], M, ns) * 0.7 + 0.3
wout = torch.randn(B, chain[-1], M)
ref = copy.deepcopy(mlp).double().train()
acts = []
h = x0.double().requires_grad_(True)
xin = h
for mod in ref:
    h = mod(h)
    h.retain_grad()
    acts.append(h)
(h.max(dim=3)[0] * wout.double()).sum().backward()
n = len(chain) - 1
Yr = [acts[3 * k] for k in range(n)]          # conv outputs
Ar = [acts[3 * k + 2] for k in range(n)]      # relu outputs
def rel(a, b):
    return float((a.cpu().double() - b).abs().max() / max(1e-300, float(b.abs().max())))
m2 = copy.deepcopy(mlp).to(dev).train()
x = x0.to(dev)
count = B * M * ns
flag = fused._overflow_flag(dev)
ys, ps, ws, was = [], [], [], []
operand, pin, mode = x, None, ext.TIN_RAW
convs = [m2[3 * k] for k in range(n)]; bns = [m2[3 * k + 1] for k in range(n)]
for k in range(n):
    w = convs[k].weight.detach().reshape(chain[k + 1], chain[k]).contiguous()
    y = torch.empty((B, chain[k + 1], M, ns), device=dev)
    wa = w.abs().amax().reshape(1)
    partial = ext.tconv(w, wa, mode, ext.TEPI_STATS, y, operand=operand, pin=pin, overflow=flag)
    P = torch.empty((chain[k + 1], 8), device=dev)
    ext.tbn_finalize(partial, count, bns[k], P)
    print(f"Y{k+1} {rel(y, Yr[k].detach()):.1e}", end="  ")
    ys.append(y); ps.append(P); ws.append(w); was.append(wa)
    operand, pin, mode = y, P, ext.TIN_BNRELU
out, arg, yarg = ext.tpool_fwd(ys[-1], ps[-1])
print(f"out {rel(out, Ar[-1].detach().max(dim=3)[0]):.1e}")
gout = wout.to(dev).contiguous()
amax = torch.zeros((n,), device=dev)
dg, db = ext.tbn_bwd_finalize(ext.tpool_bwd_stats(yarg, gout, ps[-1], amax_out=amax[n - 1:]), count, ps[-1])
print(f"dgamma{n} {rel(dg, ref[3*n-2].weight.grad):.1e} dbeta{n} {rel(db, ref[3*n-2].bias.grad):.1e}")
dA = None
for k in range(n - 1, -1, -1):
    pool = dict(gout=gout, arg=arg, nsample=ns) if k == n - 1 else dict(dA=dA)
    dw = ext.twgrad(ys[k], ps[k], ys[k - 1] if k else x, ps[k - 1] if k else None, amax[k:], overflow=flag, **pool)
    print(f"dW{k+1} {rel(dw, ref[3*k].weight.grad.reshape(dw.shape)):.1e}", end="  ")
    din = dict(operand=dA) if k < n - 1 else dict(gout=gout, arg=arg, nsample=ns)
    md = ext.TIN_BNBWD if k < n - 1 else ext.TIN_BNBWD_POOL
    if k > 0:
        prev = torch.empty_like(ys[k - 1])
        partial = ext.tconv(ws[k], was[k], md, ext.TEPI_BWD, prev, y=ys[k], pin=ps[k], epi_y=ys[k - 1], pout=ps[k - 1], transposed=True, overflow=flag, amax_in=amax[k:], amax_out=amax[k - 1:], **din)
        dg, db = ext.tbn_bwd_finalize(partial, count, ps[k - 1])
        print(f"dA{k} {rel(prev, Ar[k-1].grad):.1e} dgamma{k} {rel(dg, ref[3*k-2].weight.grad):.1e} dbeta{k} {rel(db, ref[3*k-2].bias.grad):.1e}")
        # where is dA wrong?
        e = (prev.cpu().double() - Ar[k - 1].grad).abs()
        if float(e.max()) > 1e-4 * float(Ar[k - 1].grad.abs().max()):
            bad = (e > 1e-4 * float(Ar[k - 1].grad.abs().max())).nonzero()
            print("   wrong elements:", bad.shape[0], "first", bad[:3].tolist(), "last", bad[-3:].tolist(),
                  "scenes", sorted(set(bad[:, 0].tolist())), "rows", (int(bad[:, 1].min()), int(bad[:, 1].max())),
                  "cols", (int((bad[:, 2] * ns + bad[:, 3]).min()), int((bad[:, 2] * ns + bad[:, 3]).max())))
        dA = prev
    else:
        dx = torch.empty_like(x)
        ext.tconv(ws[0], was[0], md, ext.TEPI_NONE, dx, y=ys[0], pin=ps[0], transposed=True, overflow=flag, amax_in=amax[0:], **din)
        print(f"dx {rel(dx, xin.grad):.1e}")
torch.cuda.synchronize()
print("overflow", fused.check_overflow())
