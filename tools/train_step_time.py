"""Time a TRAINING step (forward + backward, BatchNorm on batch statistics) of the IA-SSD SA layers 0-2 through the
op-by-op path (HIP sampling / query / group kernels + the train-mode conv / BatchNorm kernels).
usage: python tools/train_step_time.py [B] [N] [reps] [ab]
  ab: alternate sa_stack.STREAM_TRAINING_QUERIES on / off inside one process (three rounds each) and print the medians
  abs: the same for pointnet2_modules.SCALES_ON_STREAMS (a layer's scales on streams of their own)
  abp: the same with sa_stack.prefetch_first_layer for the next batch issued before every backward
  abf: the same for pointnet2_modules.FUSED_MLP_TRAINING (csrc/mlp_train.hip against the op-by-op kernels)
  abc: the same for pointnet2_utils.GROUP_CONCAT_TRAINING (grouping with gradients as one launch each way)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spsnet_amd import pointnet2_modules as M, sa_stack, scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ab = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] in ("ab", "abf", "abs", "abp", "abc") else ""
PREFETCH = False
dev = torch.device("cuda:0")
layers = sa_stack.build_sa_layers(M, sa_stack.IASSD_KITTI, seed=3).to(dev).train()
xyz, feats = scenes.make_batch("kitti-lidar-v1", B, N, seed0=1)
x = torch.from_numpy(xyz).to(dev)
f = torch.from_numpy(feats).to(dev)


def step():
    for p in layers.parameters():
        p.grad = None
    outs = sa_stack.run_sa_layers(layers, x, f)
    loss = sum(o[1].square().mean() for o in outs) + sum(o[2].square().mean() for o in outs if o[2] is not None)
    if PREFETCH:      # a loop that holds its next batch (here: the same tensor): its sampling runs beside this backward
        sa_stack.prefetch_first_layer(layers, x)
    loss.backward()


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(3):
    step()
if ab:
    res = {True: [], False: []}
    for rnd in range(3):
        for flag in (True, False):
            if ab == "ab":
                sa_stack.STREAM_TRAINING_QUERIES = flag
            elif ab == "abs":
                M.SCALES_ON_STREAMS = flag
            elif ab == "abc":
                from spsnet_amd import pointnet2_utils as _U
                _U.GROUP_CONCAT_TRAINING = flag
            elif ab == "abp":
                PREFETCH = flag
                layers[0]._presampled = layers[0]._preball = None
            else:
                M.FUSED_MLP_TRAINING = flag
            step()
            res[flag].append(timed(reps))
    for flag in (True, False):
        print({"ab": "STREAM_TRAINING_QUERIES", "abs": "SCALES_ON_STREAMS", "abf": "FUSED_MLP_TRAINING", "abp": "prefetch_first_layer", "abc": "GROUP_CONCAT_TRAINING"}[ab] + f"={flag}: " + " ".join(f"{v:.2f}" for v in res[flag]) +
              f"  median {statistics.median(res[flag]):.2f} ms", flush=True)
else:
    ms = timed(reps)
    print(f"training step (SA L0-L2, fwd+bwd) {B}x{N}: {ms:.2f} ms ({B * N / ms / 1e3:.2f} M points/s)", flush=True)
