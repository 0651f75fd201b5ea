"""Kernel names of ONE training step of a backbone mirror (forward + backward, train() mode), from torch.profiler's device
activity records -- what tests/test_parity_gpu.py::test_backbone_training_step_runs asserts on: no library GEMM / convolution /
BatchNorm kernel (`Cijk_*`, `igemm_*`, `MIOpen*`, `batched_transpose*`, `SubTensorOp*`, `gemm*`) in the step.
usage: python tools/train_step_kernels.py [iassd|pagnet] [fp32|fp16x2] [B] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spsnet_amd import backbones as BB, fused, scenes

from tests.gpu_util import LIBRARY_KERNEL_MARKS as LIBRARY, kernel_names


def one_step(net, batch):
    out = net(batch)
    loss = out["centers_features"].square().mean() + out["ctr_offsets"][:, 1:].square().mean()
    for t in out["sa_ins_preds"]:
        if isinstance(t, torch.Tensor):
            loss = loss + t[..., 1:].square().mean()
    for p in net.parameters():
        p.grad = None
    loss.backward()


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "iassd"
    if len(sys.argv) > 2:
        fused.set_train_precision(sys.argv[2])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    dev = torch.device("cuda:0")
    full = N >= 16384
    base = BB.IASSD_KITTI_CFG if tag == "iassd" else BB.SPSNET_KITTI_CFG
    cfg = base if full else BB.scaled_cfg(base, [1024, 256, 128, 64, -1, 64])
    cls = BB.IASSD_Backbone if tag == "iassd" else BB.PAGNet_Backbone
    net = scenes.fill_parameters(cls(cfg, num_class=3, input_channels=4), 2).to(dev).train()
    xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=77)
    bidx = np.repeat(np.arange(B, dtype=np.float32), N)[:, None]
    points = torch.from_numpy(np.concatenate([bidx, xyz.reshape(-1, 3), feats.transpose(0, 2, 1).reshape(-1, 1)], 1).astype(np.float32)).to(dev)
    batch = dict(batch_size=B, points=points)
    if tag == "pagnet":
        batch["stds"] = torch.from_numpy(np.random.default_rng(1).uniform(0, 40, (B, N)).astype(np.float32)).to(dev)
    names = kernel_names(lambda: one_step(net, dict(batch)))
    print(f"{tag} {fused.TRAIN_PRECISION} B={B} N={N}: {sum(names.values())} launches, {len(names)} distinct kernels")
    for k, v in sorted(names.items(), key=lambda kv: -kv[1]):
        mark = "LIBRARY " if any(s in k for s in LIBRARY) else "        "
        print(f"{mark}{v:4d} x {k[:150]}")
    lib = [k for k in names if any(s in k for s in LIBRARY)]
    print("library kernels:", len(lib))
