"""Set-abstraction (SA) modules of IA-SSD / SPSNet on top of the gfx950 ops.

Class names, constructor keywords, forward signatures, return tuples and state_dict keys
(`mlps.{i}.{0,1,3,4,6,7}`, `aggregation_layer.*`, `confidence_layers.*`, `mlp_modules.*`,
`ctr_reg.*`, `mlp.*`) follow the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py so that IASSD_backbone.py:57-79,
PAGNet_backbone.py and stability_generate/model.py:84-95 can build and load checkpoints
unchanged.  The implementation is organised differently: layer stacks come from two small
builders, the sampling strategies live in a dispatch table (`_SAMPLERS`, same precedence as the
reference's if/elif chain at :284-419), score-based sampling is one fused kernel
(`sps_score_topk`) and grouping uses the fused query+group kernel when no gradient is needed.
"""
import ctypes
import os
from typing import Callable, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import fused as _fused
from . import pointnet2_batch_cuda as _ext
from . import pointnet2_utils
from . import streams


# ------------------------------------------------------------------------------------------
# builders
# ------------------------------------------------------------------------------------------
def _conv_bn_relu_stack(widths: List[int], conv, norm) -> nn.Sequential:
    """[conv1x1(bias=False), norm, ReLU] per consecutive width pair -> Sequential indices 0,1,2,3,4,5,..."""
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        layers += [conv(cin, cout, kernel_size=1, bias=False), norm(cout), nn.ReLU()]
    return nn.Sequential(*layers)


class _MaxOverSamples(torch.autograd.Function):
    """F.max_pool2d(x, [1, ns]).squeeze(-1) on (B,C,M,ns) with the same gradient routing (first maximum of a row):
    csrc/group_gather.hip sps_pool_max_fwd / _bwd instead of torch's generic pooling kernels."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, M, ns = x.shape
        out = torch.empty((B, C, M), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, C, M), dtype=torch.uint8, device=x.device)
        _ext.pool_max_fwd(x, out, arg)
        ctx.save_for_backward(arg)
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        grad_in = torch.empty(tuple(arg.shape) + (ctx.ns,), dtype=torch.float32, device=grad_out.device)
        _ext.pool_max_bwd(grad_out.contiguous(), arg, grad_in)
        return grad_in


class _BatchNormReLUTrain(torch.autograd.Function):
    """relu(batch_norm(x)) on batch statistics (training), forward and backward in csrc/bn_relu_train.hip; updates the
    module's running statistics like nn.BatchNorm2d.forward does."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        x = x.contiguous()
        y, mean, invstd = _ext.bn_relu_train_fwd(x, weight.detach(), bias.detach(), bn.eps, bn.momentum,
                                                 bn.running_mean, bn.running_var)
        bn.num_batches_tracked.add_(1)
        ctx.save_for_backward(x, mean, invstd, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, weight, bias = ctx.saved_tensors
        dx, dweight, dbias = _ext.bn_relu_train_bwd(x, dy.contiguous(), mean, invstd, weight.detach(), bias.detach())
        return dx, dweight, dbias, None


class _Conv1x1Train(torch.autograd.Function):
    """Conv2d / Conv1d(kernel 1, no bias) on (B, C, ...) with gradients, channel-major throughout: csrc/conv1x1_train.hip
    (forward, data and weight gradient in exact fp32 on the matrix cores) at EVERY width -- no library GEMM, none of MIOpen's
    NHWC implicit-GEMM kernels and their transposes.  (Round 4 sent layers of >= 64 channels to torch.matmul / bmm; the fused
    train-mode kernels serve those in exact fp32 now, and this op-by-op form is what is left for stacks they decline.)"""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        w2 = weight.detach().reshape(weight.shape[0], weight.shape[1])
        ctx.save_for_backward(x, weight)
        return _ext.conv1x1_apply(x, w2, False)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        w2 = weight.detach().reshape(weight.shape[0], weight.shape[1])
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _ext.conv1x1_apply(dy, w2, True)
        if ctx.needs_input_grad[1]:
            dw = _ext.conv1x1_wgrad(x, dy).view_as(weight)
        return dx, dw


def _L_twgrad_ws(b, co, ci, l):
    return _lib.load().sps_twgrad_workspace_floats(b, co, ci, l)


# the forward / the backward of a fused train-mode grouped MLP as ONE C call each (sps_mlp_train_forward / _backward) instead of
# ~8 / ~14 ctypes launches with a torch.empty per output: a backbone's training step is host-bound
ONE_CALL_TRAINING = os.environ.get("SPS_ONE_CALL_TRAINING", "1") != "0"
FUSED_CONV_TRAINING = True
# [Conv2d 1x1, BatchNorm2d(batch statistics), ReLU] x n + max-pool as the fused kernels of csrc/mlp_train.hip, in the
# arithmetic fused.TRAIN_PRECISION names (exact fp32 by default, split-fp16 opt-in)
FUSED_MLP_TRAINING = os.environ.get("SPS_FUSED_MLP_TRAINING", "1") != "0"
# the aggregation / confidence / vote stacks ([Conv1d, BatchNorm1d, ReLU] on (B, C, M), then for the heads a Conv1d with a bias)
# through the same kernels.  Round 4 measured the launch-by-launch form 2 % slower than torch's (MIOpen implicit-GEMM + its
# BatchNorm) at the IA-SSD shapes -- these tensors are 2-8 MB and the stack is launch-bound either way -- and left it off;
# round 5 runs it as ONE C call per stack and direction like the grouped MLPs (sps_mlp_train_forward with nsample = 0) and
# turns it on: no convolution or BatchNorm of a training step goes through a library any more.
FUSED_POINTWISE_TRAINING = os.environ.get("SPS_FUSED_POINTWISE_TRAINING", "1") != "0"
# PointnetFPModule in training: its stack runs on the fused train-mode kernels from this many input elements on (below, the
# stack is launch-bound either way, as the aggregation stacks are)
FUSED_FP_TRAINING_MIN = int(os.environ.get("SPS_FUSED_FP_TRAINING_MIN", str(1 << 21)))
# training: the scales of a layer on streams of their own (see _group_mlp_pool) -- measured SLOWER (8.72 vs 8.29 ms per step:
# every fork / join is a cross-stream dependency that is actually waited for, ~12 us each, and the big kernels of both
# chains are memory-bound and only share the bandwidth); kept as a switch, off
SCALES_ON_STREAMS = os.environ.get("SPS_SCALES_ON_STREAMS", "0") != "0"
# inference, exact fp32: the smaller grouping scale of a layer on a second stream behind the larger one (see _group_mlp_pool)
FILL_WITH_SMALL_SCALE = os.environ.get("SPS_FILL_WITH_SMALL_SCALE", "1") != "0"
# inference, exact fp32: a layer behind a streamed D-FPS layer starts its grouping on the early picks (begin_early_pool)
EARLY_POOL = os.environ.get("SPS_EARLY_POOL", "1") != "0"


def _sync_group(bn):
    """(process group, world size) when `bn` is a SyncBatchNorm that really synchronises in this process, else (None, 1)"""
    if not isinstance(bn, nn.SyncBatchNorm):
        return None, 1
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, 1
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    return (group, world) if world > 1 or _SYNC_ALWAYS else (None, 1)


_SYNC_ALWAYS = False      # tests: take the synchronised code path in a one-process group too


def _all_reduce_sums(partial, group, count=None):
    """(parts, C, 2) local partial sums -> ((1, C, 2) sums over ALL ranks of `group`, (C, 2) local sums[, global count]).
    count: this rank's element count; it rides in the same all-reduce (one more float64) and comes back as the GLOBAL count,
    a device scalar -- ranks may hold different numbers of elements (torch's SyncBatchNorm gathers per-rank counts too)."""
    import torch.distributed as dist
    local = partial.sum(dim=0)
    if count is None:
        total = local.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        return total.unsqueeze(0).contiguous(), local
    buf = torch.cat([local.reshape(-1), local.new_full((1,), float(count))])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:-1].view(1, *local.shape), local, buf[-1:]


class _GroupedMLPPoolTrain(torch.autograd.Function):
    """pool=True: max_s relu(bn_n(conv_n(... relu(bn_1(conv_1(x)))))) for a (B, C0, M, ns) grouped tensor in train() mode ->
    (B, Cn, M); pool=False: the same stack without the pool on a (B, C0, L) tensor -> (B, Cn, L) (the aggregation /
    confidence layers).  Forward and backward on csrc/mlp_train.hip: only the pre-BatchNorm convolution outputs (forward)
    and the gradients w.r.t. the post-ReLU activations (backward) are materialised; BatchNorm / ReLU / the pool's gradient
    routing / the BatchNorm backward are applied where the operands are loaded, the batch statistics come out of the
    convolutions' epilogues.  Updates the running statistics like nn.BatchNorm.forward.  Reference op sequence:
    pointnet2_modules.py:432-444 (grouped MLP + pool), :449-455 (aggregation / confidence)."""

    @staticmethod
    def forward(ctx, bns, pool, x, *wgb):
        ctx.f32 = _fused.TRAIN_PRECISION == "fp32"
        _lib.load().sps_set_train_precision(int(ctx.f32))
        with _ext.launch_scope(x):
            return _GroupedMLPPoolTrain._forward(ctx, bns, pool, x, *wgb)

    @staticmethod
    def backward(ctx, gout):
        _lib.load().sps_set_train_precision(int(ctx.f32))      # (the arithmetic its forward ran in)
        with _ext.launch_scope(gout):
            return _GroupedMLPPoolTrain._backward(ctx, gout)

    @staticmethod
    def _one_call(bns, pool, x):
        """The whole forward / backward as ONE C call each (sps_mlp_train_forward / _backward) -- what a plain (unsynchronised)
        stack takes, with its pool (a grouped MLP) or without (an aggregation / confidence / vote stack: nsample = 0 in the
        descriptor); SyncBatchNorm keeps the launch-by-launch form (its all-reduces sit between the launches)."""
        return (ONE_CALL_TRAINING and 1 <= len(bns) <= 4 and x.dim() == (4 if pool else 3)
                and all(_sync_group(bn)[0] is None for bn in bns))

    @staticmethod
    def _forward_one_call(ctx, bns, pool, x, *wgb):
        n = len(bns)
        x = x.contiguous()
        if pool:
            B, c0, M, ns = x.shape
        else:
            (B, c0, M), ns = x.shape, 0
        tail = (M, ns) if pool else (M,)
        l = M * max(ns, 1)
        dev = x.device
        ws = [wgb[3 * k].detach().reshape(wgb[3 * k].shape[0], wgb[3 * k].shape[1]).contiguous() for k in range(n)]
        cs = [c0] + [w.shape[0] for w in ws]
        d = _lib.MlpTrainDesc()
        d.n, d.b, d.m, d.ns = n, B, M, ns
        ys, ps = [], []
        for k in range(n + 1):
            d.c[k] = cs[k]
        # one arena for the small float buffers: parameter blocks, wamax
        small = torch.empty((sum(cs[1:]) * _ext.TRAIN_PARAMS + 4,), dtype=torch.float32, device=dev)
        off = 0
        for k, bn in enumerate(bns):
            y = torch.empty((B, cs[k + 1]) + tail, dtype=torch.float32, device=dev)
            p = small[off:off + cs[k + 1] * _ext.TRAIN_PARAMS].view(cs[k + 1], _ext.TRAIN_PARAMS)
            off += cs[k + 1] * _ext.TRAIN_PARAMS
            ys.append(y); ps.append(p)
            d.w[k], d.y[k], d.params[k] = ws[k].data_ptr(), y.data_ptr(), p.data_ptr()
            d.gamma[k], d.beta[k] = wgb[3 * k + 1].data_ptr(), wgb[3 * k + 2].data_ptr()
            d.eps[k], d.momentum[k] = float(bn.eps), float(bn.momentum)
            d.running_mean[k], d.running_var[k] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            d.num_batches_tracked[k] = bn.num_batches_tracked.data_ptr()
        wamax = small[off:off + 4]
        out = torch.empty((B, cs[n], M), dtype=torch.float32, device=dev)
        if pool:
            yarg = torch.empty((B, cs[n], M), dtype=torch.float32, device=dev)
            arg = torch.empty((B, cs[n], M), dtype=torch.uint8, device=dev)
        else:
            yarg = arg = out.new_empty((0,))
        need = max([B * cs[n] * 2] + [_ext.tconv_parts(B, l, c) * c * 2 for c in cs])
        partial = torch.empty((need,), dtype=torch.float64, device=dev)
        d.x, d.wamax, d.partial = x.data_ptr(), wamax.data_ptr(), partial.data_ptr()
        d.out = out.data_ptr()
        if pool:
            d.arg, d.yarg = arg.data_ptr(), yarg.data_ptr()
        d.overflow = _fused._overflow_flag(dev).data_ptr()
        _lib.check(_lib.load().sps_mlp_train_forward(ctypes.byref(d), ctypes.c_void_p(_ext._stream(x))), "mlp_train_forward")
        for bn in bns:
            _ext._bump_versions(bn.running_mean, bn.running_var, bn.num_batches_tracked)
        ctx.n, ctx.ns, ctx.count, ctx.pool = n, ns, B * l, pool
        ctx.one_call = True
        ctx.cs = cs
        ctx.wshapes = [tuple(wgb[3 * k].shape) for k in range(n)]
        ctx.save_for_backward(x, arg, yarg, small, *ys, *ws)
        return out

    @staticmethod
    def _backward_one_call(ctx, gout):
        n, ns, cs = ctx.n, ctx.ns, ctx.cs
        saved = ctx.saved_tensors
        x, arg, yarg, small = saved[0], saved[1], saved[2], saved[3]
        ys, ws = saved[4:4 + n], saved[4 + n:4 + 2 * n]
        gout = gout.contiguous()
        B, c0, M = x.shape[:3]
        l = M * max(ns, 1)
        tail = (M, ns) if ctx.pool else (M,)
        dev = x.device
        d = _lib.MlpTrainDesc()
        d.n, d.b, d.m, d.ns = n, B, M, ns
        for k in range(n + 1):
            d.c[k] = cs[k]
        # the small outputs in one arena: d gamma / d beta per layer, amax
        sg = torch.empty((2 * sum(cs[1:]) + 4,), dtype=torch.float32, device=dev)
        grads = [None] * (3 * n)
        off = po = 0
        keep = []
        work_need = 0
        for k in range(n):
            c = cs[k + 1]
            d.w[k], d.y[k] = ws[k].data_ptr(), ys[k].data_ptr()
            d.params[k] = small[po:po + c * _ext.TRAIN_PARAMS].data_ptr()
            po += c * _ext.TRAIN_PARAMS
            dg, db = sg[off:off + c], sg[off + c:off + 2 * c]
            off += 2 * c
            d.dgamma[k], d.dbeta[k] = dg.data_ptr(), db.data_ptr()
            grads[3 * k + 1], grads[3 * k + 2] = dg, db
            if ctx.needs_input_grad[3 + 3 * k]:
                dw = torch.empty((c, cs[k]), dtype=torch.float32, device=dev)
                d.dw[k] = dw.data_ptr()
                grads[3 * k] = dw.view(ctx.wshapes[k])
                work_need = max(work_need, int(_L_twgrad_ws(B, c, cs[k], l)))
            if k > 0:
                dA = torch.empty((B, cs[k]) + tail, dtype=torch.float32, device=dev)
                d.dA[k] = dA.data_ptr()
                keep.append(dA)
        dx = None
        if ctx.needs_input_grad[2]:
            dx = torch.empty_like(x)
            d.dA[0] = dx.data_ptr()
        d.wamax = small[po:po + 4].data_ptr()
        d.amax = sg[off:off + 4].data_ptr()
        need = max([B * cs[n] * 2] + [_ext.tconv_parts(B, l, c) * c * 2 for c in cs])
        partial = torch.empty((need,), dtype=torch.float64, device=dev)
        work = torch.empty((max(work_need, 1),), dtype=torch.float32, device=dev)
        d.x, d.partial, d.work = x.data_ptr(), partial.data_ptr(), work.data_ptr()
        d.gout = gout.data_ptr()
        if ctx.pool:
            d.arg, d.yarg = arg.data_ptr(), yarg.data_ptr()
        d.overflow = _fused._overflow_flag(dev).data_ptr()
        _lib.check(_lib.load().sps_mlp_train_backward(ctypes.byref(d), ctypes.c_void_p(_ext._stream(gout))), "mlp_train_backward")
        return (None, None, dx) + tuple(grads)

    @staticmethod
    def _forward(ctx, bns, pool, x, *wgb):
        if _GroupedMLPPoolTrain._one_call(bns, pool, x):
            return _GroupedMLPPoolTrain._forward_one_call(ctx, bns, pool, x, *wgb)
        ctx.one_call = False
        n = len(bns)
        x = x.contiguous()
        B, tail = x.shape[0], tuple(x.shape[2:])
        ns = tail[1] if pool else 1
        count = x.numel() // x.shape[1]
        flag = _fused._overflow_flag(x.device)
        ys, ps, syncs = [], [], []
        ws = [wgb[3 * k].detach().reshape(wgb[3 * k].shape[0], wgb[3 * k].shape[1]).contiguous() for k in range(n)]
        wamax = _ext.weights_amax(ws)          # (the kernels scale the weights by a power of two derived from it)
        was = [wamax[k:k + 1] for k in range(n)]
        operand, pin, mode = x, None, _ext.TIN_RAW
        for k, bn in enumerate(bns):
            w = ws[k]
            y = torch.empty((B, w.shape[0]) + tail, dtype=torch.float32, device=x.device)
            partial = _ext.tconv(w, was[k], mode, _ext.TEPI_STATS, y, operand=operand, pin=pin, overflow=flag)
            params = torch.empty((w.shape[0], _ext.TRAIN_PARAMS), dtype=torch.float32, device=x.device)
            group, world = _sync_group(bn)
            gcount = None
            if group is not None:        # SyncBatchNorm: the statistics AND the element count of the whole global batch
                partial, _, gcount = _all_reduce_sums(partial, group, count)
            _ext.tbn_finalize(partial, count, bn, params, count_dev=gcount)   # (also counts the batch: num_batches_tracked += 1)
            syncs.append((group, gcount))
            ys.append(y); ps.append(params)
            operand, pin, mode = y, params, _ext.TIN_BNRELU
        if pool:
            out, arg, yarg = _ext.tpool_fwd(ys[-1], ps[-1])
        else:
            out = _ext.tbn_apply_relu(ys[-1], ps[-1])
            arg = yarg = out.new_empty((0,))
        ctx.n, ctx.ns, ctx.count, ctx.pool = n, ns, count, pool
        ctx.wshapes = [tuple(wgb[3 * k].shape) for k in range(n)]
        ctx.syncs = syncs
        ctx.save_for_backward(x, arg, yarg, *ys, *ps, *ws, *was)
        return out

    @staticmethod
    def _backward(ctx, gout):
        if ctx.one_call:
            return _GroupedMLPPoolTrain._backward_one_call(ctx, gout)
        n, ns, count, pool = ctx.n, ctx.ns, ctx.count, ctx.pool
        saved = ctx.saved_tensors
        x, arg, yarg = saved[0], saved[1], saved[2]
        ys, ps, ws, was = saved[3:3 + n], saved[3 + n:3 + 2 * n], saved[3 + 2 * n:3 + 3 * n], saved[3 + 3 * n:3 + 4 * n]
        gout = gout.contiguous()
        flag = _fused._overflow_flag(x.device)
        grads = [None] * (3 * n)
        # amax[k]: largest magnitude of the gradient that enters layer k's BatchNorm backward (the kernels scale their fp16
        # operands by an exact power of two derived from it: csrc/mlp_train.hip)
        amax = torch.zeros((n,), dtype=torch.float32, device=x.device)
        # BatchNorm-backward sums of the last layer: from the pooled gradient alone (its dA is never materialised), or from
        # the dense incoming gradient of a stack without a pool
        last = (_ext.tpool_bwd_stats(yarg, gout, ps[-1], amax_out=amax[n - 1:]) if pool
                else _ext.tbn_bwd_stats(ys[-1], gout, ps[-1], amax_out=amax[n - 1:]))
        def finalize_bwd(partial, k):
            """BatchNorm-backward sums of layer k -> ps[k][:, 6:8] (means over the GLOBAL batch under SyncBatchNorm) and the
            (local) gradients of its weight and bias, as torch's SyncBatchNorm returns them"""
            group, gcount = ctx.syncs[k]
            if group is None:
                return _ext.tbn_bwd_finalize(partial, count, ps[k])
            total, local = _all_reduce_sums(partial, group)
            _ext.tbn_bwd_finalize(total, count, ps[k], count_dev=gcount)     # (the forward's all-reduced global count)
            return local[:, 1].float(), local[:, 0].float()

        grads[3 * n - 2], grads[3 * n - 1] = finalize_bwd(last, n - 1)
        dA, dx = (None if pool else gout), None
        for k in range(n - 1, -1, -1):
            routed = pool and k == n - 1            # the incoming gradient is the pooled one, routed by the arg-max in the load
            if ctx.needs_input_grad[3 + 3 * k]:
                src = dict(gout=gout, arg=arg, nsample=ns) if routed else dict(dA=dA)
                dw = _ext.twgrad(ys[k], ps[k], ys[k - 1] if k else x, ps[k - 1] if k else None, amax[k:], overflow=flag, **src)
                grads[3 * k] = dw.view(ctx.wshapes[k])
            din = dict(gout=gout, arg=arg, nsample=ns) if routed else dict(operand=dA)
            mode = _ext.TIN_BNBWD_POOL if routed else _ext.TIN_BNBWD
            if k > 0:
                prev = torch.empty_like(ys[k - 1])
                partial = _ext.tconv(ws[k], was[k], mode, _ext.TEPI_BWD, prev, y=ys[k], pin=ps[k], epi_y=ys[k - 1], pout=ps[k - 1],
                                     transposed=True, overflow=flag, amax_in=amax[k:], amax_out=amax[k - 1:], **din)
                grads[3 * k - 2], grads[3 * k - 1] = finalize_bwd(partial, k - 1)
                dA = prev
            elif ctx.needs_input_grad[2]:
                dx = torch.empty_like(x)
                _ext.tconv(ws[0], was[0], mode, _ext.TEPI_NONE, dx, y=ys[0], pin=ps[0], transposed=True, overflow=flag, amax_in=amax[0:],
                           **din)
        return (None, None, dx) + tuple(grads)


# widest layer side the fused train-mode kernels take (sps_tconv: K slabs of 256 input rows, sps_twgrad: 256 x 256 blocks of dW;
# IA-SSD's widest is layer 5's aggregation stack, 512 + 1024 = 1536 -> 512 channels, IA-SSD.yaml:55)
_FUSED_TRAIN_MAX_CHANNELS = int(os.environ.get("SPS_FUSED_TRAIN_MAX_CHANNELS", "2048"))


def _fused_stack_train(mods, x, pool: bool):
    """[conv 1x1 (no bias), BatchNorm on batch statistics, ReLU] x n (+ max over the last axis when pool) through
    _GroupedMLPPoolTrain, or None when the modules / shapes do not qualify.  mods: the flat module list."""
    # pool: Conv2d / BatchNorm2d on (B, C, M, ns); without: Conv1d / BatchNorm1d, or Conv2d 1x1 / BatchNorm2d (the
    # feature-propagation stacks, which the reference applies to (B, C, N, 1)), on a (B, C, L) tensor
    # (nn.SyncBatchNorm -- tools/train.py --sync_bn converts every BatchNorm -- takes the same kernels with its sums all-reduced)
    conv_t = nn.Conv2d if pool else (nn.Conv1d, nn.Conv2d)
    bn_t = (nn.BatchNorm2d, nn.SyncBatchNorm) if pool else (nn.BatchNorm1d, nn.BatchNorm2d, nn.SyncBatchNorm)
    if not (FUSED_MLP_TRAINING and x.is_cuda and x.dtype == torch.float32
            and x.dim() == (4 if pool else 3) and torch.is_grad_enabled()):
        return None
    if len(mods) % 3 or not mods:
        return None
    cols = x.shape[2] * (x.shape[3] if pool else 1)
    # (up to 2048 channels on either side of a layer: IA-SSD layer 5's 256 / 512 / 1024-wide scales and its 1536 -> 512
    #  aggregation stack; beyond 256 input rows sps_tconv runs K slabs and sps_twgrad 256 x 256 blocks of the weight gradient)
    if (pool and x.shape[3] not in (4, 8, 16, 32, 64)) or cols % 64 or x.shape[1] > _FUSED_TRAIN_MAX_CHANNELS or x.numel() == 0:
        return None
    bns, wgb = [], []
    for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
        if not (isinstance(conv, conv_t) and isinstance(bn, bn_t) and isinstance(act, nn.ReLU) and bn.affine
                and bn.track_running_stats and bn.momentum is not None and bn.training and all(k == 1 for k in conv.kernel_size)
                and all(v == 1 for v in conv.stride) and conv.groups == 1 and conv.bias is None
                and conv.weight.dtype == torch.float32 and bn.weight.dtype == torch.float32
                and max(conv.weight.shape[:2]) <= _FUSED_TRAIN_MAX_CHANNELS):
            return None
        bns.append(bn)
        wgb += [conv.weight, bn.weight, bn.bias]
    return _GroupedMLPPoolTrain.apply(tuple(bns), pool, x, *wgb)


def _fused_mlp_pool_train(mlp: nn.Sequential, x: torch.Tensor, pool_method: str):
    """-> pooled (B, C, M) through _GroupedMLPPoolTrain, or None when the stack / shapes do not qualify."""
    if pool_method != 'max_pool' or not mlp.training:
        return None
    return _fused_stack_train(list(mlp), x, True)


def _shared_mlp(mlp: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """mlp(x) for a [Conv2d, BatchNorm2d, ReLU]* stack; in training on the GPU each BatchNorm2d + ReLU pair runs as the
    fused batch-statistics kernels instead of MIOpen's BatchNorm and a separate ReLU (same arithmetic, 1e-6)."""
    if not (mlp.training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()):
        return mlp(x)
    mods = list(mlp)
    triples = list(zip(mods[0::3], mods[1::3], mods[2::3]))
    if len(mods) % 3 or not triples:
        return mlp(x)
    for conv, bn, act in triples:
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU) and bn.affine
                and bn.track_running_stats and bn.momentum is not None and bn.training
                and bn.weight.dtype == torch.float32):
            return mlp(x)
    for conv, bn, _ in triples:
        if (FUSED_CONV_TRAINING and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1
                and conv.bias is None and conv.weight.dtype == torch.float32 and (x.shape[2] * x.shape[3]) % 16 == 0):
            z = _Conv1x1Train.apply(x, conv.weight)
        else:
            z = conv(x)
        x = _BatchNormReLUTrain.apply(z, bn.weight, bn.bias, bn)
    return x


def _head_conv(conv: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """conv(x) for the last layer of a confidence / vote head -- Conv1d(kernel 1) WITH a bias on (B, C, M) (reference :239-243,
    :480): with gradients on the GPU the product runs on csrc/conv1x1_train.hip (forward, data and weight gradient in exact
    fp32 on the matrix cores) and the bias is one broadcast add, instead of MIOpen's implicit-GEMM kernels."""
    if (FUSED_POINTWISE_TRAINING and isinstance(conv, nn.Conv1d) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
            and torch.is_grad_enabled() and conv.kernel_size == (1,) and conv.stride == (1,) and conv.groups == 1
            and conv.padding == (0,) and conv.weight.dtype == torch.float32 and x.shape[2] % 16 == 0 and x.numel() > 0):
        y = _Conv1x1Train.apply(x, conv.weight)
        return y if conv.bias is None else y + conv.bias.view(1, -1, 1)
    return conv(x)


def _pointwise_stack(stack: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """stack(x) for an aggregation / confidence / vote stack on (B, C, M) (reference :213-245, :470-480): in training its
    leading [Conv1d, BatchNorm1d, ReLU] triples run on the fused kernels of csrc/mlp_train.hip, a class-score / offset
    convolution with its bias behind them on csrc/conv1x1_train.hip (_head_conv); anything else through torch."""
    mods = list(stack)
    lead = 0
    while (lead + 3 <= len(mods) and isinstance(mods[lead], nn.Conv1d) and isinstance(mods[lead + 1], nn.BatchNorm1d)
           and isinstance(mods[lead + 2], nn.ReLU)):
        lead += 3
    if stack.training and FUSED_POINTWISE_TRAINING and x.is_cuda and torch.is_grad_enabled():
        y = _fused_stack_train(mods[:lead], x, False) if lead else x
        if y is not None:
            for mod in mods[lead:]:
                y = _head_conv(mod, y) if isinstance(mod, nn.Conv1d) else mod(y)
            return y
    return stack(x)


def _pool_over_samples(grouped: torch.Tensor, method: str) -> torch.Tensor:
    """(B,C,M,ns) -> (B,C,M) (reference :66-75, :434-443)."""
    window = [1, grouped.size(3)]
    if method == 'max_pool':
        if grouped.is_cuda and grouped.dtype == torch.float32 and grouped.dim() == 4 and 0 < grouped.size(3) <= 255:
            return _MaxOverSamples.apply(grouped)
        pooled = F.max_pool2d(grouped, kernel_size=window)
    elif method == 'avg_pool':
        pooled = F.avg_pool2d(grouped, kernel_size=window)
    else:
        raise NotImplementedError
    return pooled.squeeze(-1)


class _PointnetSAModuleBase(nn.Module):
    """Shared plumbing: FPS + multi-scale group/MLP/pool (reference :10-81)."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'

    def calc_square_dist(self, a, b, norm=True):
        """Pairwise ||a_i - b_j||^2 as |a|^2 + |b|^2 - 2ab; a (B,n,c), b (B,m,c) -> (B,n,m).
        Reference :19-43 (`norm` only switches 2.0 vs 2, numerically identical)."""
        a_sq = (a * a).sum(dim=-1, keepdim=True)                   # (B,n,1)
        b_sq = (b * b).sum(dim=-1, keepdim=True).transpose(1, 2)   # (B,1,m)
        cross = torch.matmul(a, b.transpose(1, 2))
        return a_sq.expand(-1, -1, b.shape[1]) + b_sq.expand(-1, a.shape[1], -1) - 2.0 * cross

    def _fused_plan(self, xyz, new_xyz, features):
        """Packed weights per scale if the whole layer can run on the fused MFMA path: inference
        (no gradient wanted, BatchNorm in eval mode), max pooling, plain ball-query groupers with
        use_xyz, 3-layer MLPs whose widths the kernel supports, B*M*nsample a multiple of 32."""
        if self.training or self.pool_method != 'max_pool' or not xyz.is_cuda:
            return None
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (xyz, new_xyz, features)):
            return None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.mlps.parameters()):
            return None
        plan = []
        if (features is not None and features.is_cuda and features.dtype == torch.float32 and features.dim() == 3
                and _fused.point_major_twin(features) is None
                and any(type(g) is pointnet2_utils.QueryAndGroup and _fused.needs_point_major(mlp, g.nsample)
                        for g, mlp in zip(self.groupers, self.mlps))):
            # strict fp32 at layer 5's widths runs on the point-major kernel only: a feature tensor that did not come out of an
            # aggregation kernel (which writes the twin) gets its (B, N, C) copy here -- one transposing copy of N x C floats
            _fused.attach_point_major_twin(features)
        for grouper, mlp in zip(self.groupers, self.mlps):
            if type(grouper) is not pointnet2_utils.QueryAndGroup or not grouper.use_xyz:
                return None
            if (xyz.shape[0] * new_xyz.shape[1] * grouper.nsample) % 32 != 0:
                return None
            packed = _fused.pack_scale(mlp, grouper.nsample, point_major=_fused.point_major_twin(features) is not None,
                                       half=features is not None and features.dtype == torch.float16)
            if packed is None:
                return None
            plan.append(packed)
        return plan

    def _layer1_per_point(self, xyz_c, new_c, feats_c, plan):
        """Layer 1's feature product once per point for every scale of the layer (fused.layer1_per_point), on the scale
        streams -- beside the ball query and the column packing the caller issues next -- or None where it does not pay.
        -> [(tensor, event, stream)] per scale."""
        B, N, M = xyz_c.shape[0], xyz_c.shape[1], new_c.shape[1]
        pre, self._hoisted = getattr(self, "_hoisted", None), None
        if pre is not None and pre[0] is feats_c and pre[1] == (B, N, M) and all(a is b for a, b in zip(pre[2], plan)):
            return pre[3]          # forward() started them before the sampler ran (_start_layer1_per_point)
        if feats_c is None or not all(_fused.can_hoist_layer1(p, B, N, M, g.nsample) for p, g in zip(plan, self.groupers)):
            return None
        # A layer that sa_stack may START on a partly written cloud (begin_early_pool: its centroids are a plain D-FPS of the
        # layer in front, the identity prefix) keeps the grouped form in BOTH schedules: its staged launches and the one-launch
        # form must pool the same bits
        types = getattr(self, "sample_type_list", None) or []
        plain_dfps = len(types) == 1 and ('D-FPS' in types[0] or 'DFS' in types[0]) and not any(
            t in types[0] for t in ('cls', 'ctr', 'ss'))
        if plain_dfps and self.early_pool_plan(xyz_c, new_c, feats_c) is not None:
            return None
        main = torch.cuda.current_stream(xyz_c.device)
        side = self._scale_streams(xyz_c)
        fork = torch.cuda.Event()
        fork.record(main)
        order = sorted(range(len(plan)), key=lambda k: -(self.groupers[k].nsample * plan[k].c3))
        res = [None] * len(plan)
        twin = _fused.point_major_twin(feats_c)
        for rank, k in enumerate(order):
            st = side[0]     # (both on the stream the small scale's launch runs on: one queue to wake, the big scale's rows first)
            rows = torch.empty((B, N, plan[k].c1), dtype=torch.float32, device=xyz_c.device)
            with torch.cuda.stream(st):
                st.wait_event(fork)
                _fused.layer1_per_point(feats_c, plan[k], out=rows)
                done = torch.cuda.Event()
                done.record(st)
            rows.record_stream(st)
            twin.record_stream(st)
            res[k] = (rows, done, st)
        return res

    def _start_layer1_per_point(self, xyz, features):
        """forward(), before the sampler runs: the per-point rows of layer 1 need the input features only, so their launches
        go out first, on the scale streams, and run beside the sampler, the ball query and the column packing."""
        self._hoisted = None
        if not (_fused.HOIST_LAYER1 and features is not None and xyz.is_cuda and not self.training and self.npoint_list
                and getattr(self, "_prepooled", None) is None and xyz.is_contiguous() and features.is_contiguous()
                and _fused.point_major_twin(features) is not None):
            return
        B, N = xyz.shape[0], xyz.shape[1]
        M = sum(min(n, N) for n in self.npoint_list if n > 0)
        shape_only = torch.empty((B, M, 3), device="meta")
        plan = self._fused_plan(xyz, shape_only, features)
        if not plan or len(plan) != len(self.groupers):
            return
        res = self._layer1_per_point(xyz, shape_only, features, plan)
        if res is not None:
            self._hoisted = (features, (B, N, M), plan, res)

    def _run_scales(self, xyz_c, new_c, feats_c, idxs, plan, out, cols, pm, hoist=None, **kw):
        """The grouped-MLP launches of a layer's scales into `out` (kw: merge / full_range_if / unless_any of
        fused.group_mlp_pool).  hoist: _layer1_per_point's result."""
        offsets = [sum(p.c3_real for p in plan[:k]) for k in range(len(plan))]

        def rows_for(k):
            """layer 1's per-point rows of scale k, ordered before the launch that is about to be issued on the current stream"""
            if hoist is None:
                return {}
            rows, done, st = hoist[k]
            cur = torch.cuda.current_stream(xyz_c.device)
            if cur.cuda_stream != st.cuda_stream:
                cur.wait_event(done)
                rows.record_stream(cur)
            return dict(hoisted=rows)

        if (len(plan) == 2 and all(c is not None for c in cols) and FILL_WITH_SMALL_SCALE
                and all(p.split == 0 and p.point_major for p in plan)):
            # The exact-fp32 kernels run one wave per SIMD, each wave walking its units: a launch ends when the waves with
            # one unit more than the others do, and for that last round half the chip idles (3640 units on 1024 waves = 4
            # rounds for 568 of them).  The launch with fewer columns goes to a second stream BEHIND the big one: its
            # workgroups land on the compute units the big launch's short workgroups free first.
            k0, k1 = sorted(range(2), key=lambda k: -(idxs[k].shape[2] * plan[k].c3))
            main = torch.cuda.current_stream(xyz_c.device)
            side = self._scale_streams(xyz_c)[0]
            rows0 = rows_for(k0)         # (its wait, if any, goes in FRONT of `ready`: the small launch must not overtake the big one)
            ready = torch.cuda.Event()
            ready.record(main)
            _fused.group_mlp_pool(xyz_c, new_c, feats_c, idxs[k0], plan[k0], out, offsets[k0], columns=cols[k0], out_point_major=pm,
                                  **rows0, **kw)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                _fused.group_mlp_pool(xyz_c, new_c, feats_c, idxs[k1], plan[k1], out, offsets[k1], columns=cols[k1],
                                      out_point_major=pm, **rows_for(k1), **kw)
                done = torch.cuda.Event()
                done.record(side)
            main.wait_event(done)
            extra = tuple(t for t in kw.values() if isinstance(t, torch.Tensor))
            for t in (xyz_c, new_c, feats_c, idxs[k1], out, cols[k1].cols, cols[k1].meta, cols[k1].ntiles) + extra + \
                    ((_fused.point_major_twin(feats_c),) if plan[k1].point_major else ()):
                if t is not None:
                    t.record_stream(side)
        else:
            for k, (idx, packed) in enumerate(zip(idxs, plan)):
                _fused.group_mlp_pool(xyz_c, new_c, feats_c, idx, packed, out, offsets[k], columns=cols[k], out_point_major=pm,
                                      **rows_for(k), **kw)

    def early_pool_plan(self, xyz, new_xyz, features):
        """The packed plan of this layer if its grouping can START on a partly written cloud (begin_early_pool), else None:
        inference on the fused path, two scales on kernels that take packed columns (all but the shared-stream split-fp16
        one), an aggregation kernel that reads point-major rows, at most 8192 centroids."""
        if len(self.groupers) != 2 or not EARLY_POOL:
            return None
        plan = self._fused_plan(xyz, new_xyz, features)
        B, M = xyz.shape[0], new_xyz.shape[1]
        # (any kernel that takes packed columns: exact fp32, split-fp16 per wave, fp16 features; not the shared-stream one)
        if not plan or any(p.split == 2 for p in plan) or B * M > 8192 or M % 4 or xyz.shape[1] < 256:
            return None
        if not all(B * M * g.nsample >= _fused.PACK_PM32_MIN_COLUMNS for g in self.groupers):
            return None
        if any(g.nsample > 32 for g in self.groupers) or not self._tail_reads_point_major(xyz, new_xyz):
            return None
        return plan

    def begin_early_pool(self, xyz, new_xyz, features, n_early, flag, flags_any):
        """Start -- or continue -- this layer's grouping while its input cloud is still being written: `xyz` / `features` (and
        the features' point-major twin) are final for the points [0, n_early) of every scene only, `new_xyz` are the layer's
        centroids AS GUESSED (sa_stack: the identity prefix of the producing D-FPS).  A ball-query row is the first nsample
        hits in index order, so the row over [0, n_early) is a prefix of the complete row, and max-pooling does not care in
        which launch a column was computed: the columns of the points that exist go through the grouped MLP now -- a second
        call with a larger n_early adds the columns of the points in between, merged by an atomic max -- and
        `_group_mlp_pool` adds the last ones when the layer is called.  flag (device int32) / flags_any (device int32 array):
        raised by then if the early inputs or the guessed centroids were wrong -- the last stage then redoes the layer from
        scratch.  Exact: same kernels, same columns, the same pooled maxima."""
        pre = getattr(self, "_prepooled", None)
        cont = pre is not None and pre[0] is xyz and pre[1] is features and pre[7] is new_xyz and pre[4] < n_early
        plan = self.early_pool_plan(xyz, new_xyz, features)
        if plan is None:
            return False
        ga, gb = self.groupers
        B, M, width = xyz.shape[0], new_xyz.shape[1], sum(p.c3_real for p in plan)
        k0 = pre[4] if cont else 0
        ia, ib = _ext.ball_query_full2_points(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, new_xyz, k0, n_early - k0)
        ca, cb, taken = _fused.pack_columns2_staged(ia, ib, False, prev=pre[3] if cont else None)
        # (zeros: the identity of the merge -- pooled values are >= 0 -- for a centroid the first stage finds no neighbour for)
        out = pre[2] if cont else torch.zeros((B, M, width), dtype=torch.float32, device=xyz.device)
        self._run_scales(xyz, new_xyz, features, (ia, ib), plan, out, [ca, cb], True, **(dict(merge=True) if cont else {}))
        self._prepooled = (xyz, features, out, taken, int(n_early), flag, flags_any, new_xyz)
        return True

    def _group_mlp_pool(self, xyz, new_xyz, features, point_major_ok=False):
        """-> pooled features (B, sum C_out, M); with point_major_ok (the caller's aggregation kernel reads either layout)
        the fused path may return them point-major, (B, M, sum C_out), tagged `_sps_point_major`: the grouped-MLP kernels
        then write a centroid's pooled rows contiguously."""
        pre, self._prepooled = getattr(self, "_prepooled", None), None
        picks, self._late_gather_idx = getattr(self, "_late_gather_idx", None), None
        plan = self._fused_plan(xyz, new_xyz, features)
        late = bool(plan and pre is not None and pre[0] is xyz and pre[1] is features and point_major_ok and
                    pre[2].shape[:2] == (xyz.shape[0], new_xyz.shape[1]) and not any(p.split == 2 for p in plan))
        if picks is not None and not (late and pre[7] is new_xyz and new_xyz.is_contiguous()):
            new_xyz.copy_(_ext.gather_xyz(xyz.contiguous(), picks.contiguous()))   # nobody will gather for us: do it here
            picks = None
        if late:
            # the early columns are in `out` already (begin_early_pool): query, pack and merge the LATE ones -- or, while a
            # repair flag is up, everything again with plain stores, inside the same three launches
            _, _, out, taken, n_early, flag, flags_any, _ = pre
            ga, gb = self.groupers
            new_c = new_xyz.contiguous()
            ia, ib = _ext.ball_query_full2_points(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, new_c, n_early,
                                                  xyz.shape[1] - n_early, full_if=flag, full_if_any=flags_any, gather_idx=picks)
            ca, cb, _ = _fused.pack_columns2_staged(ia, ib, True, prev=taken, full_if=flag, full_if_any=flags_any)
            self._run_scales(xyz, new_c, features, (ia, ib), plan, out, [ca, cb], True, merge=True, full_range_if=flag,
                             unless_any=flags_any)
            out._sps_point_major = True
            return out
        if plan:
            xyz_c, new_c = xyz.contiguous(), new_xyz.contiguous()
            feats_c = features.contiguous() if features is not None else None
            # nsample 64: a centroid's samples span several kernel units, combined by an atomic max onto zeros
            alloc = torch.zeros if any(g.nsample > 32 for g in self.groupers) else torch.empty
            B, M, width = xyz.shape[0], new_xyz.shape[1], sum(p.c3_real for p in plan)
            pack = [_fused.want_packed((B, M, g.nsample), p) for g, p in zip(self.groupers, plan)]
            # a point-major `out` is written by the packed-column and the shared-stream kernels only
            pm = bool(point_major_ok) and all(pk or p.split == 2 for pk, p in zip(pack, plan))
            out = alloc((B, M, width) if pm else (B, width, M), dtype=torch.float32, device=xyz.device)
            hoist = self._layer1_per_point(xyz_c, new_c, feats_c, plan)   # (beside the ball query and the packing below)
            if len(plan) == 2:  # the usual two-scale layer: both ball queries share one scan
                ga, gb = self.groupers
                idxs = _ext.ball_query_full2(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz_c, new_c)
            else:
                idxs = [_ext.ball_query_full(g.radius, g.nsample, xyz_c, new_c) for g in self.groupers]
            both = _fused.pack_columns2(idxs[0], idxs[1]) if (len(plan) == 2 and all(pack)) else None   # one launch for both scales
            cols = [both[k] if both is not None else (_fused.pack_columns(idx) if pk else None)
                    for k, (idx, pk) in enumerate(zip(idxs, pack))]
            self._run_scales(xyz_c, new_c, feats_c, idxs, plan, out, cols, pm, hoist=hoist)
            if pm:
                out._sps_point_major = True
            return out
        if features is not None and features.dtype == torch.float16:
            raise NotImplementedError("fp16 feature tensors are served by the fused inference path only (eval mode, no "
                                      "gradients, max pooling, IA-SSD layer widths); cast to float32 for anything else")
        scales = []
        idxs = self._neighbour_indices(xyz, new_xyz)

        def one_scale(k, grouper, mlp):
            if idxs is not None:
                grouped = pointnet2_utils.group_with_index(xyz, new_xyz, features, idxs[k], grouper.use_xyz)
            else:
                grouped = grouper(xyz, new_xyz, features)      # (B, C, M, ns)
            pooled = _fused_mlp_pool_train(mlp, grouped, self.pool_method)
            if pooled is None and self.pool_method == 'max_pool' and not mlp.training:
                pooled = _fused.generic_mlp_pool(mlp, grouped)     # inference at widths no specialised kernel serves
            return pooled if pooled is not None else _pool_over_samples(_shared_mlp(mlp, grouped), self.pool_method)

        side = self._scale_streams(xyz) if (SCALES_ON_STREAMS and self.training and xyz.is_cuda and idxs is not None
                                            and len(self.groupers) > 1 and torch.is_grad_enabled()) else None
        if side is None:
            for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
                scales.append(one_scale(k, grouper, mlp))
            return torch.cat(scales, dim=1)
        # Training: a layer's scales are independent chains of ~12 launches forward and ~20 backward, many of them small
        # (statistics finalizers, second-stage reductions): each chain on a stream of its own lets one scale's small
        # launches and tails run beside the other's big ones.  autograd replays every backward node on the stream its
        # forward ran on, so the backward splits the same way.
        main = torch.cuda.current_stream(xyz.device)
        fork = torch.cuda.Event()
        fork.record(main)
        for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            st = side[k % len(side)]
            with torch.cuda.stream(st):
                st.wait_event(fork)
                scales.append(one_scale(k, grouper, mlp))
                done = torch.cuda.Event()
                done.record(st)
            main.wait_event(done)
            scales[-1].record_stream(main)
            for t in (xyz, new_xyz, features) + tuple(idxs):
                if t is not None:
                    t.record_stream(st)
        return torch.cat(scales, dim=1)

    def _scale_streams(self, like):
        main = torch.cuda.current_stream(like.device)
        return [streams.helper(like.device, main, f"scale{i}") for i in range(2)]   # (cached per pass / role in streams.py)

    def _neighbour_indices(self, xyz, new_xyz):
        """Ball-query rows of both scales of a two-radius layer on the op-by-op (training) path, or None: the rows
        sa_stack.run_sa_layers queried while this layer's FPS was still running (`_preball`), else ONE scan of the cloud for
        both radii instead of one per scale (same rows: every row written, zeros for empty balls)."""
        pre, self._preball = getattr(self, "_preball", None), None
        if pre is not None and pre[0] is new_xyz:
            cur = torch.cuda.current_stream(new_xyz.device)
            for t in (new_xyz,) + tuple(pre[1]):      # (possibly produced on a prefetch stream: sa_stack.prefetch_first_layer)
                t.record_stream(cur)
            return pre[1]
        if not (xyz.is_cuda and len(self.groupers) == 2 and xyz.dtype == torch.float32 and xyz.is_contiguous()
                and all(type(g) is pointnet2_utils.QueryAndGroup for g in self.groupers)):
            return None
        ga, gb = self.groupers
        return _ext.ball_query_full2(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, new_xyz.detach().contiguous())

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum C_out,npoint)."""
        if new_xyz is None and self.npoint is not None:
            picked = pointnet2_utils.farthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), picked).transpose(1, 2).contiguous()
        return new_xyz, self._group_mlp_pool(xyz, new_xyz, features)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale-grouping SA layer with plain D-FPS (reference :84-125)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool', **kwargs):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3  # in place on the caller's list, like the reference (:108-109)
            self.mlps.append(_conv_bn_relu_stack(spec, nn.Conv2d, nn.BatchNorm2d))
        self.pool_method = pool_method


# ------------------------------------------------------------------------------------------
# sampling strategies.  Each takes the module and a _SampleInput and returns (B, npoint) int32;
# strategies that thin `stds` update ctx.stds.
# ------------------------------------------------------------------------------------------
class _SampleInput:
    __slots__ = ("xyz", "xyz_full", "_features", "_window", "_contig", "cls", "npoint", "stds", "whole", "new_xyz")

    def __init__(self, xyz, xyz_full, features, window, contig, cls, npoint, stds, whole=False):
        self.xyz, self.xyz_full = xyz, xyz_full
        self._features, self._window, self._contig = features, window, contig
        self.cls, self.npoint, self.stds = cls, npoint, stds
        self.whole = whole        # the window is the whole cloud: a strategy may hand back the centroid rows as well
        self.new_xyz = None       # ... here: xyz_full[idx], produced by the sampling launch itself

    @property
    def xyz_flipped(self):
        return self.xyz_full.transpose(1, 2).contiguous()

    @property
    def features_t(self):
        """(B, n_window, C) view of the features; only the feature-space samplers pay for it."""
        if self._features is None:
            return None
        ft = self._features.transpose(1, 2)[:, self._window, :]
        return ft.contiguous() if self._contig else ft


def _thin_stds(ctx: _SampleInput, idx):
    batch = ctx.xyz_full.shape[0]
    ctx.stds = pointnet2_utils.gather_operation(ctx.stds.view(batch, 1, -1).contiguous(), idx).squeeze()


def _fusable_gather(ctx):
    x = ctx.xyz_full
    return ctx.whole and x.is_cuda and x.is_contiguous() and not (torch.is_grad_enabled() and x.requires_grad)


def _sample_score(mod, ctx):
    """'cls'/'ctr' aware: top-k of sigmoid(max_c logits) (reference :287-291), fused -- with the centroid gather that
    follows it (:423-424) when the window is the whole cloud."""
    if _fusable_gather(ctx):
        idx, ctx.new_xyz = _ext.score_topk(ctx.cls.contiguous(), ctx.npoint, xyz=ctx.xyz_full)
        return idx
    return _ext.score_topk(ctx.cls.contiguous(), ctx.npoint)


def _sample_stability(mod, ctx):
    """SPSNet 'ss'/'sss': top-k of sigmoid(max_c logits) * (1 - sigmoid(stds/8 - 3)), then thin stds
    (reference :293-305), fused."""
    if ctx.stds is None:
        raise NotImplementedError
    batch, n = ctx.cls.shape[0], ctx.cls.shape[1]
    stds = ctx.stds.reshape(batch, n).contiguous()
    if _fusable_gather(ctx):
        idx, ctx.new_xyz = _ext.score_topk(ctx.cls.contiguous(), ctx.npoint, stds=stds, xyz=ctx.xyz_full)
    else:
        idx = _ext.score_topk(ctx.cls.contiguous(), ctx.npoint, stds=stds)
    _thin_stds(ctx, idx)
    return idx


def _sample_dfps(mod, ctx):
    """Distance FPS (reference :307-310).  sa_stack.run_sa_layers may have started this FPS early on a side
    stream (it depends only on the previous layer's new_xyz, not on its features): pick that result up."""
    pre = getattr(mod, "_presampled", None)
    if pre is not None:
        mod._presampled = None
        idx, done, src = pre[:3]
        if src is ctx.xyz and idx.shape[1] == ctx.npoint:
            main = torch.cuda.current_stream(idx.device)
            if done is not None:
                main.wait_event(done)
                idx.record_stream(main)
            if len(pre) > 3 and ctx.whole:
                ctx.new_xyz = pre[3]          # the streamed queries gathered the centroids already
                # ... or guessed them (sa_stack, early stages): the layer's last ball query gathers xyz[idx] into them
                mod._late_gather_idx = idx if (len(pre) > 4 and pre[4]) else None
            if ctx.stds is not None:
                _thin_stds(ctx, idx)
            return idx
    idx = pointnet2_utils.furthest_point_sample(ctx.xyz.contiguous(), ctx.npoint)
    if ctx.stds is not None:
        _thin_stds(ctx, idx)
    return idx


def _sample_sfps(mod, ctx):
    """Stability FPS: FPS seeds, then inside each seed's (ss_radii, ss_nsamples) ball pick the point
    with the smallest std; falls back to plain FPS when scene 0 keeps < 3500 distinct points
    (reference :314-353, threshold hard-coded there)."""
    if ctx.stds is None:
        raise NotImplementedError
    batch = ctx.xyz_full.shape[0]
    stds = ctx.stds.view(batch, 1, -1).contiguous()
    seeds = pointnet2_utils.furthest_point_sample(ctx.xyz.contiguous(), ctx.npoint)
    seed_xyz = pointnet2_utils.gather_operation(ctx.xyz_flipped, seeds).transpose(1, 2).contiguous()
    ball = pointnet2_utils.ball_query(mod.ss_radii, mod.ss_nsamples, ctx.xyz_full, seed_xyz)
    ball_stds = pointnet2_utils.grouping_operation(stds, ball).squeeze()
    steadiest = torch.argmin(ball_stds, dim=-1).view(ball.shape[0], -1, 1)
    idx = torch.gather(ball, 2, steadiest).view(seed_xyz.shape[0], -1)
    ctx.stds = pointnet2_utils.gather_operation(stds, idx).squeeze()
    if idx[0].unique().shape[0] < 3500:
        idx = pointnet2_utils.furthest_point_sample(ctx.xyz.contiguous(), ctx.npoint)
    return idx


def _feature_distance(mod, ctx):
    joint = torch.cat([ctx.xyz, ctx.features_t], dim=-1)
    return mod.calc_square_dist(joint, joint).contiguous()


def _sample_ffps(mod, ctx):
    """Feature FPS over xyz (+) features (reference :357-361)."""
    return pointnet2_utils.furthest_point_sample_with_dist(_feature_distance(mod, ctx), ctx.npoint)


def _sample_fs(mod, ctx):
    """Fusion sampling = F-FPS ++ D-FPS -> (B, 2*npoint) (reference :363-369)."""
    by_feature = pointnet2_utils.furthest_point_sample_with_dist(_feature_distance(mod, ctx), ctx.npoint)
    by_distance = pointnet2_utils.furthest_point_sample(ctx.xyz, ctx.npoint)
    return torch.cat([by_feature, by_distance], dim=-1)


def _sample_rand(mod, ctx):
    """One random permutation shared by the batch (reference :370-371)."""
    perm = torch.randperm(ctx.xyz.shape[1], device=ctx.xyz.device)[None, :ctx.npoint].int()
    return perm.repeat(ctx.xyz.shape[0], 1)


def _sample_partitioned(key_fn: Callable[[torch.Tensor], torch.Tensor]):
    """ds-FPS / ry-FPS: sort each scene by a scalar key, cut into 4 equal parts, FPS npoint/4 in each
    part and map back to scene indices (reference :372-419)."""
    parts = 4

    def run(mod, ctx):
        part_xyz, part_idx = [], []
        for scene in ctx.xyz:
            order = key_fn(scene).sort(dim=0, descending=False)[1]
            part_xyz.append(scene[order].view(parts, -1, 3))
            part_idx.append(order.view(parts, -1))
        part_xyz = torch.cat(part_xyz, dim=0)
        part_idx = torch.cat(part_idx, dim=0)
        local = pointnet2_utils.furthest_point_sample(part_xyz, ctx.npoint // parts)
        picked = [order[sel.long()] for sel, order in zip(local, part_idx)]
        return torch.cat(picked, dim=-1).reshape(ctx.xyz_full.shape[0], ctx.npoint).int()

    return run


# (predicate on the sample-type string, strategy); first match wins -- reference order :287-419
_SAMPLERS = [
    (lambda t: 'cls' in t or 'ctr' in t, _sample_score),
    (lambda t: 'ss' in t or 'sss' in t, _sample_stability),
    (lambda t: 'D-FPS' in t or 'DFS' in t, _sample_dfps),
    (lambda t: 'S-FPS' in t or 'SFS' in t, _sample_sfps),
    (lambda t: 'F-FPS' in t or 'FFS' in t, _sample_ffps),
    (lambda t: t == 'FS', _sample_fs),
    (lambda t: 'Rand' in t, _sample_rand),
    (lambda t: t in ('ds_FPS', 'ds-FPS'), _sample_partitioned(lambda p: p.norm(dim=-1) - 5)),
    (lambda t: t in ('ry_FPS', 'ry-FPS'), _sample_partitioned(lambda p: torch.atan(p[:, 0] / p[:, 1]))),
]


def _identity_index(batch: int, n: int, device) -> torch.Tensor:
    return torch.arange(n, device=device, dtype=torch.int32).unsqueeze(0).repeat(batch, 1)


class _SamplingSAModule(_PointnetSAModuleBase):
    """Common body of PointnetSAModuleMSG_WithSampling and PointnetSampling."""

    _allowed_samplers = None  # None = all strategies

    def _build_scales(self, npoint_list, radii, nsamples, mlps, use_xyz, dilated_group):
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint_list = npoint_list
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        width = 0
        for i, (radius, nsample, spec) in enumerate(zip(radii, nsamples, mlps)):
            if npoint_list is None:
                grouper = pointnet2_utils.GroupAll(use_xyz)
            elif dilated_group:
                inner = 0. if i == 0 else radii[i - 1]
                grouper = pointnet2_utils.QueryDilatedAndGroup(radius, inner, nsample, use_xyz=use_xyz)
            else:
                grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
            self.groupers.append(grouper)
            if use_xyz:
                spec[0] += 3  # in place on the caller's list, like the reference (:199-201)
            self.mlps.append(_conv_bn_relu_stack(spec, nn.Conv2d, nn.BatchNorm2d))
            width += spec[-1]
        return width

    def _build_aggregation(self, width, aggregation_mlp):
        if aggregation_mlp and len(self.mlps) > 0:
            self.aggregation_layer = _conv_bn_relu_stack([width] + list(aggregation_mlp), nn.Conv1d, nn.BatchNorm1d)
            return aggregation_mlp[-1]
        self.aggregation_layer = None
        return width

    def _sample(self, xyz, features, cls_features, stds):
        """-> (sampled_idx (B, sum npoint) int32, stds).  Reference :270-424 / :709-726."""
        chunks = []
        fused_xyz = None
        begin = 0
        for sample_type, sample_range, npoint in zip(self.sample_type_list, self.sample_range_list,
                                                     self.npoint_list):
            if npoint <= 0:
                continue
            if sample_range == -1:
                window = slice(begin, None)
            else:
                window = slice(begin, sample_range)
                begin += sample_range  # sic: the reference adds the range end, not its length (:282)
            whole = window.start == 0 and window.stop is None
            xyz_w = xyz if whole else xyz[:, window, :]
            xyz_w = xyz_w if sample_range == -1 else xyz_w.contiguous()
            cls_w = cls_features[:, window, :] if cls_features is not None else None

            if xyz_w.shape[1] <= npoint:
                chunks.append(_identity_index(xyz_w.shape[0], xyz_w.shape[1], xyz_w.device))
                continue
            for accepts, strategy in _SAMPLERS:
                if accepts(sample_type) and (self._allowed_samplers is None or strategy in self._allowed_samplers):
                    ctx = _SampleInput(xyz_w, xyz, features, window, sample_range == -1, cls_w, npoint, stds, whole)
                    chunks.append(strategy(self, ctx))
                    stds = ctx.stds
                    fused_xyz = ctx.new_xyz
                    break
            else:
                raise NotImplementedError(f"sampling method {sample_type!r}")
        sampled_idx = chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=-1)
        if len(chunks) == 1 and fused_xyz is not None:
            new_xyz = fused_xyz                                     # the sampler's launch gathered the centroids already
        elif xyz.is_cuda and not (torch.is_grad_enabled() and xyz.requires_grad):
            new_xyz = _ext.gather_xyz(xyz.contiguous(), sampled_idx.contiguous())  # (B,N,3) rows, no transposes
        else:
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(),
                                                       sampled_idx).transpose(1, 2).contiguous()
        hook = getattr(self, "_on_new_xyz", None)
        if hook is not None:
            hook(new_xyz)
        return sampled_idx, new_xyz, stds

    def _tail_reads_point_major(self, xyz, new_xyz):
        """True if _tail will run the fused aggregation kernel (which reads pooled features in either layout)."""
        agg = self.aggregation_layer
        if agg is None or agg.training or not xyz.is_cuda:
            return False
        head = getattr(self, "confidence_layers", None)
        if (head is not None and head.training) or new_xyz.shape[1] % 16:
            return False
        if torch.is_grad_enabled() and any(p.requires_grad for p in agg.parameters()):
            return False
        return _fused._tail_layers(agg, head) is not None

    def _tail(self, pooled, half_out=False):
        """Aggregation stack and confidence head (reference :449-455) -> (new_features, cls (B,M,K) | None).  In
        inference both run as ONE kernel (csrc/pw_mlp.hip, BatchNorm folded, exact fp32) when their shapes allow.
        half_out: the layer's input features were fp16, so its output features are stored as fp16 too."""
        head = getattr(self, "confidence_layers", None)
        x_pm = bool(getattr(pooled, "_sps_point_major", False))
        if self.aggregation_layer is not None:
            done = _fused.pointwise_tail(self.aggregation_layer, head, pooled, half_out, x_pm)
            if done is not None:
                return done
            if x_pm:   # the fused kernel declined after all: back to the reference layout
                pooled = pooled.transpose(1, 2).contiguous()
            if half_out:
                raise NotImplementedError("fp16 feature tensors need the fused aggregation kernel (widths multiples of 16)")
            pooled = _pointwise_stack(self.aggregation_layer, pooled)
        else:
            if x_pm:
                pooled = pooled.transpose(1, 2).contiguous()
            if half_out:
                pooled = pooled.half()
        cls = _pointwise_stack(head, pooled).transpose(1, 2) if head is not None else None
        return pooled, cls

    def _abstract(self, xyz, new_xyz, features, sampled_idx):
        """-> (new_features, cls | None)"""
        if len(self.groupers) > 0:
            return self._tail(self._group_mlp_pool(xyz, new_xyz, features, self._tail_reads_point_major(xyz, new_xyz)),
                              features is not None and features.dtype == torch.float16)
        new_features = pointnet2_utils.gather_operation(features, sampled_idx).contiguous()
        head = getattr(self, "confidence_layers", None)
        return new_features, (head(new_features).transpose(1, 2) if head is not None else None)


class PointnetSAModuleMSG_WithSampling(_SamplingSAModule):
    """SA layer with a configurable down-sampler, multi-scale grouping, an aggregation MLP and an
    optional per-point confidence head (reference :128-460).

    forward(xyz (B,N,3), features (B,C,N), cls_features (B,N,K)|None, new_xyz=None, ctr_xyz=None,
            stds=...) -> (new_xyz (B,M,3), new_features (B,C',M), cls_features (B,M,K)|None,
                          sampled_idx_list (B,M) int32 | [] when ctr_xyz is given, stds)
    """

    def __init__(self, *, npoint_list: List[int], sample_range_list: List[int], sample_type_list: List[str],
                 radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 dilated_group=False, pool_method='max_pool', aggregation_mlp: List[int],
                 confidence_mlp: List[int], num_class, **kwargs):
        super().__init__()
        self.sample_type_list = sample_type_list
        self.sample_range_list = sample_range_list
        self.dilated_group = dilated_group
        ss_radii = kwargs.get('ss_radii', None)
        if ss_radii is not None and len(ss_radii) > 0:  # stable sampling ball (reference :166-173)
            self.ss_radii = ss_radii[0]
            self.ss_nsamples = kwargs['ss_nsamples'][0]

        width = self._build_scales(npoint_list, radii, nsamples, mlps, use_xyz, dilated_group)
        self.pool_method = pool_method
        width = self._build_aggregation(width, aggregation_mlp)

        if confidence_mlp:
            head = list(_conv_bn_relu_stack([width] + list(confidence_mlp), nn.Conv1d, nn.BatchNorm1d))
            head.append(nn.Conv1d(confidence_mlp[-1], num_class, kernel_size=1, bias=True))
            self.confidence_layers = nn.Sequential(*head)
        else:
            self.confidence_layers = None

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, cls_features: torch.Tensor = None,
                new_xyz=None, ctr_xyz=None, **kwargs):
        stds = kwargs.get('stds', None)
        if stds is not None:
            stds = stds.view(xyz.shape[0], 1, -1).contiguous()
        sampled_idx_list = []
        try:
            if ctr_xyz is None:
                self._start_layer1_per_point(xyz, features)
                sampled_idx_list, new_xyz, stds = self._sample(xyz, features, cls_features, stds)
            else:
                new_xyz = ctr_xyz
            new_features, cls_features = self._abstract(xyz, new_xyz, features, sampled_idx_list)
        except BaseException:
            _drop_schedule_hints(self, everything=True)
            raise
        finally:
            _drop_schedule_hints(self)
        return new_xyz, new_features, cls_features, sampled_idx_list, stds


def _drop_schedule_hints(mod, everything=False):
    """The schedules (sa_stack, backbones, this module's own forward) hand a layer results they started early as attributes
    keyed by tensor IDENTITY: `_hoisted` (layer 1's per-point rows, started before the sampler), `_prepooled` /
    `_late_gather_idx` (columns pooled while the cloud was still being written), `_presampled` / `_preball` (an early-started
    sampler / ball query).  Each is consumed -- read and cleared -- by the forward it was made for; the reference's modules
    keep no such state ("otherwise stateless").  This runs at the END of every forward: what belongs to the
    forward that just ran is dropped whether or not a path consumed it, and after a forward that RAISED everything is
    (nothing stays pinned on the module; a later forward recomputes).  Threading contract: INTEGRATION.md."""
    mod._hoisted = None
    mod._late_gather_idx = None
    mod._prepooled = None
    if everything:        # (a normal exit keeps a sampling result prefetched for a FUTURE forward: sa_stack.prefetch_first_layer)
        mod._presampled = None
        mod._preball = None


class Vote_layer(nn.Module):
    """Light voting module with a clamped centre offset (reference :462-516).

    forward(xyz (B,M,3), features (B,C,M)) -> (vote_xyz, new_features (B,M,0), xyz, ctr_offsets)
    """

    def __init__(self, mlp_list, pre_channel, max_translate_range):
        super().__init__()
        self.mlp_list = mlp_list
        if len(mlp_list) > 0:
            # NOTE: the reference rebuilds its layer list inside the loop, so with more than one entry
            # only the last block would survive; every shipped config has exactly one entry.
            self.mlp_modules = _conv_bn_relu_stack([pre_channel] + list(mlp_list), nn.Conv1d, nn.BatchNorm1d)
            pre_channel = mlp_list[-1]
        else:
            self.mlp_modules = None
        self.ctr_reg = nn.Conv1d(pre_channel, 3, kernel_size=1)
        self.max_offset_limit = (torch.tensor(max_translate_range).float()
                                 if max_translate_range is not None else None)

    def _limit_on(self, device):
        cached = getattr(self, '_limit_cache', None)
        if cached is None or cached[0].device != device:
            limit = self.max_offset_limit.to(device).view(1, 1, 3)
            cached = (limit, -limit)         # (both bounds once: no negation launch per forward)
            object.__setattr__(self, '_limit_cache', cached)
        return cached

    def forward(self, xyz, features, **kwargs):
        if kwargs.get('center_surface_futures', None) is not None:
            self.center_surface_futures = kwargs['center_surface_futures']
        surface = getattr(self, 'center_surface_futures', None) if self.mlp_modules is not None else None
        offsets = None
        if self.mlp_modules is not None:   # inference: regression stack as one kernel (fused.vote_offsets)
            offsets = _fused.vote_offsets(self.mlp_modules, self.ctr_reg, [features] if surface is None else [surface, features])
        if offsets is None:
            hidden = features
            if self.mlp_modules is not None:
                if surface is not None:
                    hidden = torch.cat([surface, hidden], dim=1)
                hidden = _pointwise_stack(self.mlp_modules, hidden.contiguous())
            offsets = (_head_conv(self.ctr_reg, hidden) if self.training else self.ctr_reg(hidden)).transpose(1, 2)   # (B, M, 3)
        new_features = offsets[..., 3:]                      # empty: ctr_reg has exactly 3 outputs
        ctr_offsets = offsets[..., :3]
        if self.max_offset_limit is not None:
            limit, lower = self._limit_on(xyz.device)
            # where(o > l, l, o) then where(. < -l, -l, .) of the reference (:505-507) in one op; NaN stays NaN either way
            vote_xyz = xyz + torch.clamp(ctr_offsets, min=lower, max=limit)
        else:
            vote_xyz = xyz + ctr_offsets
        return vote_xyz, new_features, xyz, ctr_offsets


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale SA layer (reference :519-536)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz, pool_method=pool_method)


class PointnetFPModule(nn.Module):
    """Feature propagation by inverse-distance 3-NN interpolation (reference :539-587)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = _conv_bn_relu_stack(list(mlp), nn.Conv2d, nn.BatchNorm2d)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor, neighbours=None, point_major_ok=False) -> torch.Tensor:
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B,mlp[-1],n).
        neighbours: (dist, idx) of three_nn(unknown, known) when the caller already has them (the search needs coordinates
        only, so a backbone can run it beside the encoder: backbones.PointNet2MSG).
        point_major_ok: the caller wants per-point rows and takes (B,n,mlp[-1]) tagged `_sps_point_major` when the fused
        inference kernel serves the module (it then writes them directly; anything else returns the usual layout)."""
        if known is not None:
            dist, idx = neighbours if neighbours is not None else pointnet2_utils.three_nn(unknown, known)
            # inference: one kernel, the weights (reference :572-574) formed inside it
            fused = _fused.fp_module_mlp(self.mlp, known_feats, unknow_feats, idx, None, dist=dist, point_major=point_major_ok)
            if fused is not None:
                return fused
            inv = 1.0 / (dist + 1e-8)
            weight = inv / torch.sum(inv, dim=2, keepdim=True)
            spread = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            spread = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        stacked = torch.cat([spread, unknow_feats], dim=1) if unknow_feats is not None else spread
        if self.mlp.training and stacked.numel() >= FUSED_FP_TRAINING_MIN:
            # training at the resolution of a fine level: the stack on the fused train-mode kernels (no pool behind it)
            out = _fused_stack_train(list(self.mlp), stacked.contiguous(), False)
            if out is not None:
                return out
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)


class PointnetSampling(_SamplingSAModule):
    """D-FPS-only SA layer used by the stability generator (reference :590-763).

    forward(...) -> (new_xyz (B,M,3), new_features (B,C',M), sampled_idx_list (B,M) int32)
    """

    _allowed_samplers = (_sample_dfps,)

    def __init__(self, *, npoint_list: List[int], sample_range_list: List[int], sample_type_list: List[str],
                 radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 dilated_group=False, pool_method='max_pool', aggregation_mlp: List[int]):
        super().__init__()
        self.sample_type_list = sample_type_list
        self.sample_range_list = sample_range_list
        self.dilated_group = dilated_group
        width = self._build_scales(npoint_list, radii, nsamples, mlps, use_xyz, dilated_group)
        self.pool_method = pool_method
        self._build_aggregation(width, aggregation_mlp)
        self.confidence_layers = None

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, cls_features: torch.Tensor = None,
                new_xyz=None, ctr_xyz=None, **kwargs):
        sampled_idx_list = []
        try:
            if ctr_xyz is None:
                sampled_idx_list, new_xyz, _ = self._sample(xyz, features, cls_features, None)
            else:
                new_xyz = ctr_xyz
            new_features, _ = self._abstract(xyz, new_xyz, features, sampled_idx_list)
        except BaseException:
            _drop_schedule_hints(self, everything=True)
            raise
        finally:
            _drop_schedule_hints(self)
        return new_xyz, new_features, sampled_idx_list
