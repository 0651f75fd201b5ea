"""Host side of the fused inference path of one SA layer (eval mode, BatchNorm folded).

`sps_sa_group_mlp` (csrc/sa_mlp.hip) replaces, per grouping scale, the reference's
grouping_operation x2 + cat + [Conv2d 1x1 + BatchNorm2d + ReLU] x3 + max_pool2d
(pointnet2_modules.py:429-447).  This module folds the BatchNorm statistics into the convolution
weights, packs them in the kernel's MFMA fragment order and caches the result on the nn.Sequential.

Fragment order (see the kernel header): layer 1 is [tile][k-step][lane]; layers 2 and 3 are
[tile][k-step/4][lane][4], with k-step (t, r) of a later layer covering input channels 16t + 4q + r
(q = lane >> 4) -- the order in which the previous layer's accumulators sit in the lanes.
"""
import torch
import torch.nn as nn

from . import _lib

_L = _lib.load()


def _pad16(c):
    return (c + 15) // 16 * 16


def _fold(conv, bn):
    """Conv(1x1, no bias) + BatchNorm(eval) -> (W', b') with y = W' x + b'."""
    w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if conv.bias is not None:
        bias = bias + conv.bias.detach().float() * scale
    return w * scale[:, None], bias


def _pack_first(w, cout_pad):
    cout, cin = w.shape
    ks = (cin + 3) // 4
    wp = w.new_zeros(cout_pad, 4 * ks)
    wp[:cout, :cin] = w
    # [tile, i, ks, q] -> [tile, ks, q, i]; lane = q*16 + i
    return wp.view(cout_pad // 16, 16, ks, 4).permute(0, 2, 3, 1).contiguous().view(-1)


def _pack_next(w, cout_pad, cin_pad):
    cout, cin = w.shape
    wp = w.new_zeros(cout_pad, cin_pad)
    wp[:cout, :cin] = w
    # [tile, i, t, q, r] -> [tile, t, q, i, r]; one dwordx4 per lane = k-steps (t, 0..3)
    return wp.view(cout_pad // 16, 16, cin_pad // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def _pack_first_pm(w, cout_pad, c_feat):
    """Layer 1 of the exact-fp32 kernel for point-major features (csrc/sa_mlp_pm.hip): w (cout, 3 + c_feat) in the reference's
    channel order [x, y, z, features] -> [coordinate fragments [tile][lane = 16 q + i]: slot q < 3 = axis q, slot 3 = 0]
    followed by the feature part packed like a chained layer (k-step (t, r) <-> feature channel 16 t + 4 q + r)."""
    cout = w.shape[0]
    wx = w.new_zeros(cout_pad, 4)
    wx[:cout, :3] = w[:, :3]
    coord = wx.view(cout_pad // 16, 16, 4).permute(0, 2, 1).contiguous().view(-1)
    return torch.cat([coord, _pack_next(w[:, 3:], cout_pad, c_feat)])


def _split_halves(frag_f32, lanes_inner=4):
    """fp32 fragment tensor [..., lane, 4] -> int16-bit tensor [..., lane, hi x4 | lo x4] (the split-fp16 kernel's
    16-byte-per-lane weight fragment)."""
    hi = frag_f32.half()
    lo = (frag_f32 - hi.float()).half()
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.int16).view(-1)


def _frags_k32(w, cout_pad, cin_pad):
    """(cout, cin) fp32 -> split-fp16 K = 32 fragments as int16 [tile, k32, hi|lo, lane = 16 q + i, 8 halves]:
    k-slot (q, j) of step s is input channel 32 s + 16 (j // 4) + 4 q + j % 4, row i of the tile."""
    cout, cin_ = w.shape
    wp = w.new_zeros(cout_pad, cin_pad)
    wp[:cout, :cin_] = w
    # channel = 32 s + 16 h + 4 q + r  ->  [tile, i, s, h, q, r] -> [tile, s, q, i, h, r] -> [tile, s, lane, j]
    f = wp.view(cout_pad // 16, 16, cin_pad // 32, 2, 4, 4).permute(0, 2, 4, 1, 3, 5)
    f = f.reshape(cout_pad // 16, cin_pad // 32, 64, 8)
    hi = f.half()
    lo = (f - hi.float()).half()
    return torch.stack([hi, lo], dim=2).contiguous().view(torch.int16)


def _pack_f16(w, cout_pad, cin_pad):
    """[tile][k32] fragments of csrc/sa_mlp_f16.hip (2 KiB each)."""
    return _frags_k32(w, cout_pad, cin_pad).reshape(-1)


def _pack_h16(w, cout_pad, cin_pad):
    """[tile][k32] fragments of the pure-fp16 mode of csrc/sa_mlp_f16.hip (1 KiB each): the weights rounded to fp16 once,
    same k-slot order as _frags_k32."""
    cout, cin_ = w.shape
    wp = w.new_zeros(cout_pad, cin_pad)
    wp[:cout, :cin_] = w
    f = wp.view(cout_pad // 16, 16, cin_pad // 32, 2, 4, 4).permute(0, 2, 4, 1, 3, 5)
    return f.reshape(cout_pad // 16, cin_pad // 32, 64, 8).half().contiguous().view(torch.int16).reshape(-1)


def _pad32(c):
    return (c + 31) // 32 * 32


def _pack_stream(w1, w2, w3, c1, c2, c3, cin):
    """The three layers' split-fp16 fragments as ONE stream in the order csrc/sa_mlp_f16_lds.hip consumes them:
    layer 1 as [k32][tile]; then, for every k32-step s of layer 3, the layer-2 tiles 2s and 2s+1 as [tile][k32]
    followed by the layer-3 fragments (tile, s) of all tiles.  A fragment is 2 KiB: [lane][8 hi halves] then
    [lane][8 lo halves]; k-slot (q, j) of step s is input channel 32 s + 16 (j // 4) + 4 q + j % 4 (lane = 16 q + i)."""
    cin_pad = (cin + 31) // 32 * 32
    f1, f2, f3 = _frags_k32(w1, c1, cin_pad), _frags_k32(w2, c2, c1), _frags_k32(w3, c3, c2)
    parts = [f1.permute(1, 0, 2, 3, 4).reshape(-1)]
    for s in range(c2 // 32):
        parts.append(f2[2 * s:2 * s + 2].reshape(-1))
        parts.append(f3[:, s].reshape(-1))
    return torch.cat(parts).contiguous()


def _lds_stream_ok(c1, c2, c3):
    # measured (profiles/round1): the 64-wide scales are bound by their per-unit gathers, where four lockstep waves per
    # workgroup lose to independent ones; the shared stream pays from 128 channels on
    return (c1, c2, c3) in ((128, 128, 256), (128, 256, 256), (256, 256, 512), (256, 512, 1024))


def _pad_bias(b, cpad):
    out = b.new_zeros(cpad)
    out[:b.numel()] = b
    return out


class PackedScale:
    __slots__ = ("c1", "c2", "c3", "c3_real", "cin", "w1", "b1", "w2", "b2", "w3", "b3", "key", "split", "point_major", "half")


def _stack_layers(mlp):
    """[(conv, bn)] x 3 if `mlp` is exactly [Conv2d 1x1, BatchNorm2d, ReLU] x 3, else None."""
    mods = list(mlp)
    if len(mods) != 9:
        return None
    pairs = []
    for k in range(0, 9, 3):
        conv, bn, act = mods[k], mods[k + 1], mods[k + 2]
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU)):
            return None
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1 or not bn.track_running_stats:
            return None
        pairs.append((conv, bn))
    return pairs


def _version_key(pairs, device):
    """What a folded / packed inference cache of these (conv, bn) pairs depends on.  The running statistics' own versions are
    NOT enough: torch's native batch_norm updates running_mean / running_var inside the kernel without touching their version
    counters (and so did this library's train-mode kernels, which bump them now), so an eval() forward after train-mode
    forwards with frozen weights (BatchNorm recalibration, swa_utils.update_bn, lr = 0) would reuse stale folds.  Every
    train-mode forward of a BatchNorm module does move `num_batches_tracked` (nn.BatchNorm.forward: add_(1)); it is part of
    the key."""
    vs = [device]
    for conv, bn in pairs:
        for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, getattr(bn, "num_batches_tracked", None)):
            if t is not None:
                vs.append((t.data_ptr(), t._version))
    return tuple(vs)


# Grouped-MLP arithmetic of the INFERENCE path:
# "fp32" (default): every scale on the exact fp32 MFMA kernels (v_mfma_f32_16x16x4_f32) -- the reference's arithmetic
#           (fp32 Conv2d / BatchNorm2d / ReLU, pointnet2_modules.py:429-447); what bench.py's headline measures.
# "fp16x2" (opt-in): scales whose first hidden width is >= 32 run on the fp16 matrix cores with every fp32 operand
#           carried as a hi+lo pair of halves (3 MFMAs per block, fp32 accumulate; ~22 significant bits, <= 2e-5 relative,
#           csrc/sa_mlp_f16.hip); narrower scales stay on the fp32 MFMA kernel (their 4-channel input would be padded to 16).
PRECISION = __import__("os").environ.get("SPS_MLP_PRECISION", "fp32")
if PRECISION not in ("fp32", "fp16x2"):
    raise ValueError(f"SPS_MLP_PRECISION={PRECISION!r}: expected fp32 or fp16x2")
# Arithmetic of the fused TRAIN-mode kernels (csrc/mlp_train.hip), independent of the inference switch above:
# "fp32" (default, round 5): exact fp32 on v_mfma_f32_16x16x4_f32 -- the reference's arithmetic (fp32 Conv2d / BatchNorm2d on
#           batch statistics, pointnet2_modules.py:203-211, 432-444), same fused kernels, nothing scaled or split;
# "fp16x2" (opt-in): split-fp16 MFMA with exact power-of-two operand scaling (gradients within 1-3e-6 of float64 torch) --
#           the GEMMs of a training step then stay memory-bound instead of bound by the fp32 matrix pipe.
TRAIN_PRECISION = __import__("os").environ.get("SPS_TRAIN_PRECISION", "fp32")
if TRAIN_PRECISION not in ("fp32", "fp16x2"):
    raise ValueError(f"SPS_TRAIN_PRECISION={TRAIN_PRECISION!r}: expected fp32 or fp16x2")
# wide split-fp16 scales: the four waves of a workgroup share one weight stream through LDS (csrc/sa_mlp_f16_lds.hip)
SHARE_WEIGHTS = True
# exact-fp32 scales whose input features carry a point-major twin run on csrc/sa_mlp_pm.hip (one 16-byte load per four
# channels, a unit's inputs requested one unit ahead) instead of the channel-major gathers of csrc/sa_mlp.hip
PM_FP32 = __import__("os").environ.get("SPS_PM_FP32", "1") != "0"
_OVERFLOW = {}


def set_precision(mode):
    """Select the grouped-MLP arithmetic of the inference path ("fp32" or "fp16x2"); returns the previous mode."""
    global PRECISION
    if mode not in ("fp32", "fp16x2"):
        raise ValueError(mode)
    old, PRECISION = PRECISION, mode
    return old


def set_train_precision(mode):
    """Select the arithmetic of the fused train-mode kernels ("fp32" = exact, the reference's, the default; "fp16x2" = split-fp16
    operands); returns the previous mode.  A forward records its mode and its backward runs in the same one."""
    global TRAIN_PRECISION
    if mode not in ("fp32", "fp16x2"):
        raise ValueError(mode)
    old, TRAIN_PRECISION = TRAIN_PRECISION, mode
    return old


def _overflow_flag(device):
    key = (device.type, device.index)
    if key not in _OVERFLOW:
        _OVERFLOW[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return _OVERFLOW[key]


def check_overflow():
    """True if a split-fp16 launch met an operand beyond the exactly splittable range (|x| > 65 504) since the last
    call; such values were clamped.  Synchronises the device.  Use set_precision("fp32") for unbounded inputs."""
    bad = False
    for f in _OVERFLOW.values():
        bad |= bool(int(f.item()))
        f.zero_()
    return bad


def pack_scale(mlp, nsample, point_major=False, half=False):
    """-> PackedScale (cached on the module) or None when the fused kernel has no variant for this scale.
    point_major: layer 1's input channels ordered [features, xyz] for gathers from a (B, N, C) feature tensor
    (only used, and only honoured, by the fp16 kernels with C % 4 == 0).
    half: the feature tensors are fp16 (BASELINE configs[4]): weights rounded to fp16, one MFMA per product block
    (mode 3 of sps_sa_group_mlp_ex), hidden widths padded to multiples of 32 with zero rows."""
    pairs = _stack_layers(mlp)
    if pairs is None:
        return None
    (c1m, _), (c2m, _), (c3m, _) = pairs
    pad = _pad32 if half else _pad16
    c1, c2, c3 = pad(c1m.out_channels), pad(c2m.out_channels), _pad16(c3m.out_channels)
    # "wide" scales (IA-SSD layer 5: 256 / 512 / 1024 channels): served by the shared-stream split-fp16 kernel (mode 2) and, in
    # strict fp32, by the point-major kernel (sa_mlp_pm.hip) -- never by the channel-major per-wave kernel (sa_mlp.hip)
    wide = bool(_L.sps_sa_group_mlp_supported_stream(c1, c2, c3, nsample)) and not half
    device = c1m.weight.device
    c_feat = c1m.in_channels - 3
    pm32 = bool(point_major and not half and PM_FP32 and PRECISION == "fp32" and c_feat % 16 == 0
                and _L.sps_sa_group_mlp_pm_supported(c_feat, c1, c2, c3, nsample))   # exact fp32 on the twin (sa_mlp_pm.hip)
    stream_only = wide and not pm32
    if stream_only and not (PRECISION == "fp16x2" and SHARE_WEIGHTS):
        return None
    if not wide and not _L.sps_sa_group_mlp_supported(c1, c2, nsample):
        return None
    if not wide and not pm32 and c3 > 256:
        return None      # (the per-wave kernels stage at most 256 last-layer biases in LDS; the op-by-op path serves the scale)
    point_major = pm32 or bool(point_major and (half or PRECISION == "fp16x2") and c1 >= 32 and c_feat >= 4 and c_feat % 4 == 0)
    key = _version_key(pairs, device) + (PRECISION, SHARE_WEIGHTS, point_major, half, pm32)
    slot = ("_sps_packed_pm" if point_major else "_sps_packed") + ("_h" if half else "")
    cached = getattr(mlp, slot, None)
    if cached is not None and cached.key == key:
        return cached
    with torch.no_grad():
        (w1, b1), (w2, b2), (w3, b3) = (_fold(c, b) for c, b in pairs)
        if point_major and not pm32:
            w1 = torch.cat([w1[:, 3:], w1[:, :3]], dim=1)  # grouped channels as [features, xyz]
        p = PackedScale()
        p.point_major, p.half = point_major, bool(half)
        p.c1, p.c2, p.c3, p.c3_real, p.cin = c1, c2, c3, c3m.out_channels, c1m.in_channels
        p.split = 0
        if half:
            p.split = 3
        elif PRECISION == "fp16x2" and c1 >= 32:
            p.split = 2 if (SHARE_WEIGHTS and _lds_stream_ok(c1, c2, c3)) else 1
        cin_pad = (c1m.in_channels + 31) // 32 * 32
        if p.split == 3:
            p.w1, p.w2, p.w3 = _pack_h16(w1, c1, cin_pad), _pack_h16(w2, c2, c1), _pack_h16(w3, c3, c2)
        elif p.split == 2:
            p.w1 = _pack_stream(w1, w2, w3, c1, c2, c3, c1m.in_channels)
            p.w2 = p.w3 = p.w1
        elif p.split:
            p.w1, p.w2, p.w3 = _pack_f16(w1, c1, cin_pad), _pack_f16(w2, c2, c1), _pack_f16(w3, c3, c2)
        elif pm32:
            p.w1, p.w2, p.w3 = _pack_first_pm(w1, c1, c_feat), _pack_next(w2, c2, c1), _pack_next(w3, c3, c2)
        else:
            p.w1, p.w2, p.w3 = _pack_first(w1, c1), _pack_next(w2, c2, c1), _pack_next(w3, c3, c2)
        p.b1, p.b2, p.b3 = _pad_bias(b1, c1), _pad_bias(b2, c2), _pad_bias(b3, c3)
        p.key = key
    object.__setattr__(mlp, slot, p)  # plain attribute: not a parameter/buffer, not in state_dict
    return p


def needs_point_major(mlp, nsample):
    """True if, in the current arithmetic, only the point-major kernel serves this scale (strict fp32 at IA-SSD layer 5's
    widths): the caller then gives the feature tensor its (B, N, C) twin (attach_point_major_twin) before planning."""
    if PRECISION != "fp32" or not PM_FP32:
        return False
    pairs = _stack_layers(mlp)
    if pairs is None:
        return False
    (c1m, _), (c2m, _), (c3m, _) = pairs
    c_feat = c1m.in_channels - 3
    c1, c2, c3 = _pad16(c1m.out_channels), _pad16(c2m.out_channels), _pad16(c3m.out_channels)
    return bool(c_feat > 0 and c_feat % 16 == 0 and _L.sps_sa_group_mlp_supported_stream(c1, c2, c3, nsample)
                and _L.sps_sa_group_mlp_pm_supported(c_feat, c1, c2, c3, nsample))


def point_major_twin(features):
    """The (B, N, C) copy that pointwise_tail attached to a (B, C, N) feature tensor it produced, or None."""
    t = getattr(features, "_sps_nc", None)
    if t is None or features is None:
        return None
    B, C, N = features.shape
    if t.shape != (B, N, C) or t.device != features.device or t.dtype != features.dtype or not t.is_contiguous() or C % 4:
        return None
    return t


# Only the DISTINCT columns of a ball go through the grouped MLP (csrc/pack_columns.hip): a ball-query row repeats its first
# hit in the slots it could not fill and max-pooling is idempotent, so the pooled features are the same bit for bit.  Applied
# to whole-layer launches of at least this many columns (the pack kernel is one more launch); the shared-stream kernel
# (mode 2) keeps the padded form.
# PACK_COLUMNS: False (default) = never, True = launches of many wave rounds, "always" = wherever the kernels allow it.
# OFF by default, because on MI355X it does not pay at IA-SSD's sizes (profiles/round2, 8 x 16 384 KITTI-shaped scenes,
# un-profiled HIP-event times): a whole-layer launch of layers 1-2 is ONE to FOUR rounds of waves, so its time is set by
# a unit's latency and the round count, not by the number of units -- 0.87 x the columns of the 131-128-256-256 scale are
# still four rounds (0.318 ms padded, 0.321 ms packed, fp32), and the 6-10 us pack launch is on the critical path; the
# generator's M = N layer (many rounds, but VALU-bound narrow widths) lost 0.87 -> 0.91 ms to the packed kernels' larger
# register footprint.  Kept as a tested option (bit-identical results) for shapes with many rounds of wide units.
PACK_COLUMNS = {"1": True, "2": "always"}.get(__import__("os").environ.get("SPS_PACK_COLUMNS", "0"), False)
PACK_PM32 = __import__("os").environ.get("SPS_PACK_PM32", "1") != "0"
PACK_PM32_MIN_COLUMNS = 1 << 16
PACK_MIN_COLUMNS = 1 << 15
PACK_BIG_COLUMNS = 1 << 20
_COUNTERS = {}


def _zero_counter(device):
    """A zeroed int32 on the device without a fill launch per use: slices of a pre-zeroed buffer, refilled every 256 uses."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros((1,), dtype=torch.int32, device=device)   # (zeroed inside the graph, on every replay)
    key = (device.type, device.index, _lib.raw_stream(device))
    buf, used = _COUNTERS.get(key, (None, 256))
    if used >= 256:
        buf, used = torch.zeros((256,), dtype=torch.int32, device=device), 0
    _COUNTERS[key] = (buf, used + 1)
    return buf[used:used + 1]


class PackedColumns:
    __slots__ = ("cols", "meta", "ntiles", "cap")


def pack_columns(idx, j0=0, jcount=None):
    """idx (B, M, nsample) int32 -> PackedColumns: the tile stream of csrc/pack_columns.hip for centroids [j0, j0 + jcount)
    of every scene (one launch; the tile count stays on the device)."""
    B, M, ns = idx.shape
    jcount = M - j0 if jcount is None else jcount
    p = PackedColumns()
    p.cap = int(_L.sps_pack_columns_capacity(B, jcount, ns))
    p.cols = torch.empty((p.cap * 16,), dtype=torch.int32, device=idx.device)
    p.meta = torch.empty((p.cap * 16,), dtype=torch.int32, device=idx.device)
    p.ntiles = _zero_counter(idx.device)
    _lib.check(_L.sps_pack_columns(B, M, j0, jcount, ns, idx.data_ptr(), p.cols.data_ptr(), p.meta.data_ptr(),
                                   p.ntiles.data_ptr(), p.cap, _lib.raw_stream(idx.device)), "pack_columns")
    return p


def pack_columns2(idx_a, idx_b):
    """pack_columns for both grouping scales of a layer in ONE launch -> (PackedColumns, PackedColumns)."""
    B, M, _ = idx_a.shape
    assert idx_b.shape[:2] == (B, M)
    out = []
    for idx in (idx_a, idx_b):
        p = PackedColumns()
        p.cap = int(_L.sps_pack_columns_capacity(B, M, idx.shape[2]))
        p.cols = torch.empty((p.cap * 16,), dtype=torch.int32, device=idx.device)
        p.meta = torch.empty((p.cap * 16,), dtype=torch.int32, device=idx.device)
        p.ntiles = _zero_counter(idx.device)
        out.append(p)
    a, b = out
    _lib.check(_L.sps_pack_columns2(B, M, 0, M, idx_a.shape[2], idx_a.data_ptr(), a.cols.data_ptr(), a.meta.data_ptr(),
                                    a.ntiles.data_ptr(), a.cap, idx_b.shape[2], idx_b.data_ptr(), b.cols.data_ptr(),
                                    b.meta.data_ptr(), b.ntiles.data_ptr(), b.cap,
                                    _lib.raw_stream(idx_a.device)), "pack_columns2")
    return a, b


def pack_columns2_staged(idx_a, idx_b, last, prev=None, full_if=None, full_if_any=None):
    """One STAGE of both scales (csrc/pack_columns.hip, staged mode): idx_* = the rows of ball_query_full2_points over one point
    range (-1 rows: no hit there); prev = (counts_a, counts_b), (B, M) int32 each: the columns every centroid took in the
    stages over the lower ranges, or None for the first stage; last: this is the stage over the top range (a centroid without
    any neighbour then gets the reference's empty-ball column, point 0) -> (PackedColumns, PackedColumns, (counts_a, counts_b)
    including this stage).  Only the columns the complete rows would hold are packed; whole rows of idx_* while a repair flag
    is up."""
    B, M, _ = idx_a.shape
    out, taken = [], []
    for idx in (idx_a, idx_b):
        p = PackedColumns()
        p.cap = int(_L.sps_pack_columns_capacity(B, M, idx.shape[2]))
        p.cols = torch.empty((p.cap * 16,), dtype=torch.int32, device=idx.device)
        p.meta = torch.empty((p.cap * 16,), dtype=torch.int32, device=idx.device)
        p.ntiles = _zero_counter(idx.device)
        out.append(p)
        taken.append(torch.empty((B, M), dtype=torch.int32, device=idx.device))
    a, b = out
    ptr = lambda t: 0 if t is None else t.data_ptr()
    pa, pb = prev if prev is not None else (None, None)
    _lib.check(_L.sps_pack_columns2_late(B, M, 1 if last else 0, idx_a.shape[2], ptr(pa), idx_a.data_ptr(), taken[0].data_ptr(),
                                         a.cols.data_ptr(), a.meta.data_ptr(), a.ntiles.data_ptr(), a.cap, idx_b.shape[2],
                                         ptr(pb), idx_b.data_ptr(), taken[1].data_ptr(), b.cols.data_ptr(), b.meta.data_ptr(),
                                         b.ntiles.data_ptr(), b.cap, ptr(full_if), ptr(full_if_any),
                                         0 if full_if_any is None else full_if_any.numel(),
                                         _lib.raw_stream(idx_a.device)), "pack_columns2_late")
    return a, b, tuple(taken)


def want_packed(idx_shape, packed):
    """Pack a launch over idx of this (B, M, nsample) shape?  Whole-layer launches of the per-wave kernels where dropping
    the padded columns was measured to pay for the extra launch (see PACK_COLUMNS)."""
    B, M, ns = idx_shape
    cols = B * M * ns
    if packed.split == 2 or ns > 64 or B > 256 or M >= (1 << 20):
        return False
    if packed.split == 0 and packed.point_major and PACK_PM32 and PACK_COLUMNS != "always":
        # the exact-fp32 kernel on point-major features (sa_mlp_pm.hip) runs at ~75 % of the matrix pipe whatever the launch
        # size, so its time follows the column count: pack wherever a launch is long enough to pay for the pack launch
        return cols >= PACK_PM32_MIN_COLUMNS
    if not PACK_COLUMNS or cols < PACK_MIN_COLUMNS:
        return False
    if PACK_COLUMNS == "always":
        return True
    return cols >= PACK_BIG_COLUMNS or (packed.split == 0 and packed.c1 >= 128)


def attach_point_major_twin(features):
    """Give a (B, C, N) feature tensor that did not come from pointwise_tail its (B, N, C) twin, so that the next SA
    layer's grouped MLP takes the point-major gather path it takes inside a stack; returns `features`."""
    if features is not None and features.dim() == 3 and features.shape[1] % 4 == 0:
        features._sps_nc = features.transpose(1, 2).contiguous()
    return features


# Layer 1 of the exact-fp32 point-major kernel over the FEATURE channels once per point instead of once per grouped point
# (csrc/sa_mlp_pm.hip HOIST1): a point's features meet the same weights in every ball it falls into.  Pays when a layer has
# several times more grouped points than points (IA-SSD layer 2: 16 x); the sums are formed in a different order (features
# first, per point; then the coordinates) -- within 1e-4 of torch either way, like any two fp32 GEMM schedules.
# Measured at the bench shape (MI355X, strict fp32): the grouped launches of layer 2 get 12-18 % shorter (257 -> 227 us, 95 -> 78 us
# alone; together 256 -> 230); the two per-point launches in front of them (10 us each, on a scale stream beside the sampler, the
# ball query and the packing) give half of it back: 2.291 -> 2.277 ms per pass.  SPS_HOIST_LAYER1=0 keeps the grouped form.
HOIST_LAYER1 = __import__("os").environ.get("SPS_HOIST_LAYER1", "1") != "0"
HOIST_LAYER1_MIN_RATIO = 4


def can_hoist_layer1(packed, B, N, M, nsample):
    """Would layer 1's per-point form serve this scale (exact fp32 on the twin, supported widths, enough reuse)?"""
    if not (HOIST_LAYER1 and packed.point_major and packed.split == 0 and not packed.half):
        return False
    return bool(_L.sps_sa_layer1_per_point_supported(packed.cin - 3, packed.c1, nsample)) and M * nsample >= HOIST_LAYER1_MIN_RATIO * N


def layer1_per_point(features, packed, out=None):
    """b1 + W1f . features[point] for every point of the features' (B, N, C) twin -> (B, N, c1) fp32 (current stream)."""
    twin = point_major_twin(features)
    if twin is None:
        raise ValueError("layer1_per_point needs the feature tensor's (B, N, C) twin")
    B, N, C = twin.shape
    if out is None:
        out = torch.empty((B, N, packed.c1), dtype=torch.float32, device=twin.device)
    with torch.cuda.device(twin.device):
        _lib.check(_L.sps_sa_layer1_per_point(B * N, C, packed.c1, twin.data_ptr(), packed.w1.data_ptr(), packed.b1.data_ptr(),
                                              out.data_ptr(), _lib.raw_stream(twin.device)),
                   "sa_layer1_per_point")
    return out


def group_mlp_pool(xyz, new_xyz, features, idx, packed, out, channel_offset, j0=0, jcount=None, columns=None,
                   out_point_major=False, run_if=None, full_range_if=None, merge=False, unless_any=None, hoisted=None):
    """One launch: gather the nsample neighbours in `idx` (B,M,ns), run the packed 3-layer MLP, max-pool,
    and write channels [channel_offset, channel_offset + c3_real) of out (B, Ctot, M) -- or (B, M, Ctot) with
    out_point_major; optionally only for the centroids [j0, j0+jcount) of every scene.  A scale packed with point_major
    reads the features' (B, N, C) twin.  columns: a PackedColumns of `idx` (pack_columns): only the distinct neighbours of
    every ball are computed, same result.  run_if (device int32): the launch does nothing while it is zero;
    full_range_if (device int32): the launch covers all M centroids instead of its range while it is nonzero.
    merge (packed columns, exact fp32 on point-major features): the pooled rows are merged into `out` by an atomic max --
    `out` holds the maxima over columns an earlier launch computed -- unless full_range_if or any of unless_any (device
    int32 array) is nonzero: then plain stores (the launch covers every column: a repair)."""
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    jcount = M if jcount is None else jcount
    c_feat = 0 if features is None else features.shape[1]
    mode = int(packed.split)
    if 3 + c_feat != packed.cin:
        raise ValueError(f"grouped input has {3 + c_feat} channels, the MLP expects {packed.cin}")
    if hoisted is not None:
        # layer 1's feature product per point (layer1_per_point): the kernel gathers its rows instead of the features
        if not (packed.point_major and packed.split == 0 and hoisted.shape == (B, N, packed.c1) and hoisted.is_contiguous()):
            raise ValueError("`hoisted` is the (B, N, c1) tensor of layer1_per_point for an exact-fp32 point-major scale")
        features, c_feat = hoisted, packed.c1
        mode |= 4 | 32
    elif packed.point_major:
        twin = point_major_twin(features)
        if twin is None:
            raise ValueError("scale packed for point-major features, but the feature tensor carries no (B, N, C) twin")
        features = twin
        mode |= 4
    if out_point_major:
        mode |= 8
    if merge:
        if columns is None or packed.split == 2:
            raise ValueError("merge mode needs packed columns (served by every kernel but the shared-stream one)")
        mode |= 16
    if features is not None and (features.dtype == torch.float16) != bool(packed.half):
        raise ValueError(f"feature tensor is {features.dtype}, but the scale was packed for {'fp16' if packed.half else 'fp32'}")
    c_total = out.shape[2] if out_point_major else out.shape[1]
    stream = _lib.raw_stream(xyz.device)
    cp = (0, 0, 0, 0) if columns is None else (columns.cols.data_ptr(), columns.meta.data_ptr(), columns.ntiles.data_ptr(),
                                               columns.cap)
    _lib.check(_L.sps_sa_group_mlp_packed_merge(
        B, N, M, j0, jcount, c_feat, ns, xyz.data_ptr(), new_xyz.data_ptr(), 0 if features is None else features.data_ptr(),
        idx.data_ptr(), cp[0], cp[1], cp[2], cp[3], packed.c1, packed.c2, packed.c3, packed.c3_real, packed.w1.data_ptr(),
        packed.b1.data_ptr(), packed.w2.data_ptr(), packed.b2.data_ptr(), packed.w3.data_ptr(), packed.b3.data_ptr(),
        out.data_ptr(), c_total, channel_offset, mode,
        _overflow_flag(xyz.device).data_ptr() if packed.split else 0, 0 if run_if is None else run_if.data_ptr(),
        0 if full_range_if is None else full_range_if.data_ptr(),
        0 if unless_any is None else unless_any.data_ptr(), 0 if unless_any is None else unless_any.numel(), stream),
        "sa_group_mlp")


# ------------------------------------------------------------------ aggregation + confidence stacks (csrc/pw_mlp.hip)
def _pack_pw(w, cout_pad):
    """(cout, cin) -> [tile][k16][lane = 16 q + i][r] = W[16 tile + i][16 k16 + 4 r + q]"""
    cout, cin = w.shape
    wp = w.new_zeros(cout_pad, cin)
    wp[:cout] = w
    return wp.view(cout_pad // 16, 16, cin // 16, 4, 4).permute(0, 2, 4, 1, 3).contiguous().view(-1)


class PackedTail:
    __slots__ = ("key", "cin", "c1", "c2", "classes", "w1", "b1", "w2", "b2", "w3", "b3")


def _tail_layers(agg, head):
    """(conv, bn) of the aggregation stack and (conv, bn), conv of the head if they have the reference's shape."""
    mods = list(agg)
    if len(mods) != 3 or not (isinstance(mods[0], nn.Conv1d) and isinstance(mods[1], nn.BatchNorm1d) and isinstance(mods[2], nn.ReLU)):
        return None
    convs = [mods[0]]
    out = [(mods[0], mods[1])]
    if head is not None:
        h = list(head)
        if len(h) != 4 or not (isinstance(h[0], nn.Conv1d) and isinstance(h[1], nn.BatchNorm1d) and isinstance(h[2], nn.ReLU)
                               and isinstance(h[3], nn.Conv1d)):
            return None
        convs += [h[0], h[3]]
        out += [(h[0], h[1]), h[3]]
    for cv in convs:
        if cv.kernel_size != (1,) or cv.stride != (1,) or cv.groups != 1:
            return None
    for _, bn in out[:2] if head is not None else out[:1]:
        if not bn.track_running_stats:
            return None
    return out


class TailRunner:
    """The aggregation stack (+ confidence head) of an SA layer on `pooled` (B, C, M), launched for the whole layer or
    range by range (`run(j0, jcount)`, on torch's current stream); `result()` -> (new_features (B, Cagg, M), cls | None)."""

    def __init__(self, packed, pooled, with_head, half_out=False, x_pm=False):
        B, M = pooled.shape[0], (pooled.shape[1] if x_pm else pooled.shape[2])
        self.x_pm = bool(x_pm)           # pooled is point-major (B, M, C): what the grouped-MLP kernels write contiguously
        dev = pooled.device
        self.packed, self.x, self.B, self.M = packed, pooled, B, M
        self.half_out = bool(half_out)   # features leave as fp16 (BASELINE configs[4]); the class scores stay fp32
        ftype = torch.float16 if half_out else torch.float32
        self.y1 = torch.empty((B, packed.c1, M), dtype=ftype, device=dev)
        self.y1t = torch.empty((B, M, packed.c1), dtype=ftype, device=dev)  # point-major twin for the next SA layer
        self.y3 = torch.empty((B, M, packed.classes), dtype=torch.float32, device=dev) if with_head else None

    def tensors(self):
        return tuple(t for t in (self.y1, self.y1t, self.y3) if t is not None)

    def run(self, j0=0, jcount=None, run_if=None, full_range_if=None):
        p = self.packed
        ptr = lambda t: 0 if t is None else t.data_ptr()
        _lib.check(_L.sps_pointwise_mlp_ex(self.B, self.M, j0, self.M - j0 if jcount is None else jcount, p.cin, p.c1, p.c2,
                                           p.classes, self.x.data_ptr(), ptr(p.w1), ptr(p.b1), ptr(p.w2), ptr(p.b2),
                                           ptr(p.w3), ptr(p.b3), self.y1.data_ptr(), self.y1t.data_ptr(), ptr(self.y3),
                                           (1 if self.half_out else 0) | (2 if self.x_pm else 0),
                                           0 if run_if is None else run_if.data_ptr(),
                                           0 if full_range_if is None else full_range_if.data_ptr(),
                                           _lib.raw_stream(self.x.device)), "pointwise_mlp")

    def result(self):
        self.y1._sps_nc = self.y1t
        return self.y1, self.y3


def tail_runner(agg, head, pooled, half_out=False, x_pm=False):
    """-> TailRunner, or None when the fused path does not apply (training, gradients wanted, widths that are not
    multiples of 16, more than 16 classes, ...).  `pooled` must be contiguous and may still be being filled."""
    if agg is None or agg.training or (head is not None and head.training) or not pooled.is_cuda or pooled.dtype != torch.float32:
        return None
    if torch.is_grad_enabled() and (pooled.requires_grad or any(p.requires_grad for p in agg.parameters())):
        return None
    layers = _tail_layers(agg, head)
    if layers is None or not pooled.is_contiguous():
        return None
    (c_agg, bn_agg) = layers[0]
    B, cin, M = (pooled.shape[0], pooled.shape[2], pooled.shape[1]) if x_pm else pooled.shape
    c1 = c_agg.out_channels
    if cin != c_agg.in_channels or cin % 16 or c1 % 16 or M % 16:
        return None
    c2 = classes = 0
    if head is not None:
        (c_h, bn_h), c_out = layers[1], layers[2]
        c2, classes = c_h.out_channels, c_out.out_channels
        if c2 % 16 or classes > 16 or c_h.in_channels != c1 or c_out.in_channels != c2:
            return None
    if 17 * 4 * (max(cin, c2) + c1) > 150 * 1024:
        return None
    pairs = [(c_agg, bn_agg)] + ([layers[1]] if head is not None else [])
    key = _version_key(pairs, pooled.device)
    if head is not None:
        key = key + tuple((t.data_ptr(), t._version) for t in (layers[2].weight, layers[2].bias) if t is not None)
    packed = getattr(agg, "_sps_tail", None)
    if packed is None or packed.key != key:
        with torch.no_grad():
            packed = PackedTail()
            packed.key, packed.cin, packed.c1, packed.c2, packed.classes = key, cin, c1, c2, classes
            w, b = _fold(c_agg, bn_agg)
            packed.w1, packed.b1 = _pack_pw(w, c1), b.contiguous()
            packed.w2 = packed.b2 = packed.w3 = packed.b3 = None
            if head is not None:
                w, b = _fold(*layers[1])
                packed.w2, packed.b2 = _pack_pw(w, c2), b.contiguous()
                c_out = layers[2]
                w3 = c_out.weight.detach().reshape(classes, c2).float()
                b3 = c_out.bias.detach().float() if c_out.bias is not None else w3.new_zeros(classes)
                packed.w3, packed.b3 = _pack_pw(w3, 16), _pad_bias(b3, 16)
        object.__setattr__(agg, "_sps_tail", packed)
    return TailRunner(packed, pooled, head is not None, half_out, x_pm)


def pointwise_tail(agg, head, pooled, half_out=False, x_pm=False):
    """Aggregation stack (+ confidence head) of an SA layer on `pooled` (B, C, M) -- (B, M, C) with x_pm -- as one kernel ->
    (new_features (B, Cagg, M), cls (B, M, K) | None), or None when the fused path does not apply."""
    runner = tail_runner(agg, head, pooled.contiguous(), half_out, x_pm)
    if runner is None:
        return None
    runner.run()
    return runner.result()


class _PackedGeneric:
    __slots__ = ("key", "ws", "wamax", "params")


_GENERIC_MAX_CIN = 4096     # input channels of a layer that sps_tconv serves (beyond 288: K slabs of 256 rows, weights in LDS)


def generic_mlp_pool(mlp, grouped):
    """max over the samples of [Conv2d 1x1 + BatchNorm2d(eval) + ReLU] x n on a grouped tensor (B, C0, M, ns) for stacks NO
    specialised kernel serves (widths outside the IA-SSD / SPSNet table, any depth).  "fp16x2": the streaming convolution
    kernels of csrc/mlp_train.hip with the BatchNorm of the running statistics applied in the next layer's operand load and
    in the pool (split-fp16 MFMA, <= 2e-5 relative) -- n + 1 launches and two HBM crossings per activation instead of
    torch's 3 n + 1 launches and five; a layer with more than 288 input channels, and EVERY layer in strict "fp32", runs on
    the exact-fp32 MFMA convolution kernel (sps_conv1x1_apply) on the materialised activation.  No library GEMM / MIOpen
    convolution on the inference path in either arithmetic.  -> (B, Cn, M), or None when it does not apply (training,
    gradients wanted, nsample outside {4, 8, 16, 32, 64}, M * ns not a multiple of 64)."""
    from . import pointnet2_batch_cuda as _ext
    mods = list(mlp)
    if (len(mods) % 3 or not mods or mlp.training or not grouped.is_cuda or grouped.dtype != torch.float32
            or grouped.dim() != 4):
        return None
    exact = PRECISION == "fp32"   # strict fp32: every layer on the exact-fp32 MFMA convolution (sps_conv1x1_apply)
    B, c0, M, ns = grouped.shape
    if ns not in (4, 8, 16, 32, 64) or (M * ns) % 64 or grouped.numel() == 0:
        return None
    pairs = []
    for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU)
                and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None
                and bn.track_running_stats and bn.affine and conv.out_channels <= 1024
                and conv.weight.dtype == torch.float32):
            return None
        pairs.append((conv, bn))
    if pairs[0][0].in_channels != c0:
        return None
    if torch.is_grad_enabled() and (grouped.requires_grad or any(p.requires_grad for p in mlp.parameters())):
        return None
    key = _version_key(pairs, grouped.device)
    packed = getattr(mlp, "_sps_generic", None)
    if packed is None or packed.key != key:
        with torch.no_grad():
            packed = _PackedGeneric()
            packed.key = key
            packed.ws = [c.weight.detach().reshape(c.out_channels, c.in_channels).contiguous() for c, _ in pairs]
            packed.wamax = _ext.weights_amax(packed.ws)
            packed.params = []
            for _, bn in pairs:
                invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                scale = bn.weight.detach().float() * invstd
                shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
                zero = torch.zeros_like(scale)
                packed.params.append(torch.stack([bn.running_mean.detach().float(), invstd, scale, shift, bn.weight.detach().float(),
                                                  bn.bias.detach().float(), zero, zero], dim=1).contiguous())
        object.__setattr__(mlp, "_sps_generic", packed)
    flag = _overflow_flag(grouped.device)
    operand, pin, mode = grouped.contiguous(), None, _ext.TIN_RAW
    if not exact:
        # the streaming kernels below are shared with the training path, whose arithmetic is a process-wide C switch
        # (sps_set_train_precision; a training forward sets it for itself): this inference mode is the split-fp16 one
        _lib.load().sps_set_train_precision(0)
    with _ext.launch_scope(operand):
        for k, w in enumerate(packed.ws):
            if exact or w.shape[1] > _GENERIC_MAX_CIN:
                # strict fp32, or more input channels than the streaming kernel keeps weights for in LDS (PointRCNN's coarsest level: 515 and
                # 384): the exact-fp32 MFMA convolution of csrc/conv1x1_train.hip on the materialised activation
                a = operand if pin is None else _ext.tbn_apply_relu(operand, pin)
                y = _ext.conv1x1_apply(a, w, False)
            else:
                y = torch.empty((B, w.shape[0], M, ns), dtype=torch.float32, device=grouped.device)
                _ext.tconv(w, packed.wamax[k:k + 1], mode, _ext.TEPI_NONE, y, operand=operand, pin=pin, overflow=flag)
            operand, pin, mode = y, packed.params[k], _ext.TIN_BNRELU
        out, _, _ = _ext.tpool_fwd(operand, pin)
    return out


class _PackedFp:
    __slots__ = ("key", "c_in", "c1", "c2", "w1", "b1", "w2", "b2")


def fp_module_mlp(mlp, known_feats, unknow_feats, idx, weight, dist=None, point_major=False):
    """PointnetFPModule.forward behind its three_nn (reference :571-587): interpolation of `known_feats` (B, C2, m) with
    idx / weight (B, n, 3), concatenation with `unknow_feats` (B, C1, n) | None and the [Conv2d 1x1 + BatchNorm2d + ReLU]
    stack (one or two layers), as ONE kernel (csrc/pw_mlp.hip fp_mlp_kernel, exact fp32, BatchNorm folded) -> (B, Cout, n),
    or None when the fused path does not apply (training, gradients wanted, widths not multiples of 16, ...).
    dist (with weight None): three_nn's distances -- the weights are formed inside the kernel (the module's own three small
    torch ops otherwise); point_major: the output as per-point rows (B, n, Cout), tagged `_sps_point_major`."""
    mods = list(mlp)
    if len(mods) not in (3, 6) or mlp.training or not known_feats.is_cuda or known_feats.dtype != torch.float32:
        return None
    if weight is None:
        if dist is None or dist.dtype != torch.float32:
            return None
        weight, from_dist = dist, 1
    else:
        from_dist = 0
    pairs = []
    for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU)
                and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and bn.track_running_stats):
            return None
        pairs.append((conv, bn))
    tensors = [known_feats, weight] + ([unknow_feats] if unknow_feats is not None else [])
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or any(p.requires_grad for p in mlp.parameters())):
        return None
    B, c_known, m = known_feats.shape
    n = idx.shape[1]
    c_skip = unknow_feats.shape[1] if unknow_feats is not None else 0
    c1 = pairs[0][0].out_channels
    c2 = pairs[1][0].out_channels if len(pairs) > 1 else 0
    cin_pad = _pad16(c_known + c_skip)
    if (pairs[0][0].in_channels != c_known + c_skip or c1 % 16 or c2 % 16 or n == 0 or B > 65535
            or (unknow_feats is not None and (unknow_feats.dtype != torch.float32 or unknow_feats.shape[2] != n))
            or 17 * 4 * (max(cin_pad, c2) + (c1 if c2 else 0)) > 148 * 1024):
        return None
    key = _version_key(pairs, known_feats.device)
    packed = getattr(mlp, "_sps_fp", None)
    if packed is None or packed.key != key:
        with torch.no_grad():
            packed = _PackedFp()
            packed.key, packed.c_in, packed.c1, packed.c2 = key, cin_pad, c1, c2
            w, b = _fold(*pairs[0])
            wp = w.new_zeros(c1, cin_pad)
            wp[:, :c_known + c_skip] = w             # padded input channels read zeros and weigh nothing
            packed.w1, packed.b1 = _pack_pw(wp, c1), b.contiguous()
            packed.w2 = packed.b2 = None
            if c2:
                w2, b2 = _fold(*pairs[1])
                packed.w2, packed.b2 = _pack_pw(w2, c2), b2.contiguous()
        object.__setattr__(mlp, "_sps_fp", packed)
    cout = c2 if c2 else c1
    y = torch.empty((B, n, cout) if point_major else (B, cout, n), dtype=torch.float32, device=known_feats.device)
    kf, wt, ix = known_feats.contiguous(), weight.contiguous(), idx.contiguous()
    sk = unknow_feats.contiguous() if unknow_feats is not None else None
    _lib.check(_L.sps_fp_module_mlp_ex(B, n, m, c_known, c_skip, c1, c2, kf.data_ptr(), sk.data_ptr() if sk is not None else 0,
                                       ix.data_ptr(), wt.data_ptr(), from_dist, packed.w1.data_ptr(), packed.b1.data_ptr(),
                                       packed.w2.data_ptr() if c2 else 0, packed.b2.data_ptr() if c2 else 0, y.data_ptr(),
                                       1 if point_major else 0, _lib.raw_stream(y.device)), "fp_module_mlp")
    if point_major:
        y._sps_point_major = True
    return y


def vote_offsets(mlp, ctr_reg, parts):
    """Vote_layer's regression (pointnet2_modules.py:488-496: [Conv1d + BatchNorm1d + ReLU], Conv1d(C, 3, bias)) on the
    channel-wise concatenation of `parts` (each (B, Ci, M)) as ONE kernel -> offsets (B, M, 3), or None when the fused
    path does not apply.  Reuses the aggregation/confidence kernel (csrc/pw_mlp.hip) with an identity middle layer:
    the hidden activation is a ReLU output, so relu(I h + 0) = h exactly, in fp32."""
    mods = list(mlp) if mlp is not None else []
    if len(mods) != 3 or not (isinstance(mods[0], nn.Conv1d) and isinstance(mods[1], nn.BatchNorm1d) and isinstance(mods[2], nn.ReLU)):
        return None
    conv, bn = mods[0], mods[1]
    x0 = parts[0]
    if mlp.training or ctr_reg.training or not x0.is_cuda or any(p.dtype != torch.float32 for p in parts):
        return None
    if torch.is_grad_enabled() and (any(p.requires_grad for p in parts) or any(p.requires_grad for p in mlp.parameters())
                                    or any(p.requires_grad for p in ctr_reg.parameters())):
        return None
    if conv.kernel_size != (1,) or ctr_reg.kernel_size != (1,) or not bn.track_running_stats:
        return None
    B, M = x0.shape[0], x0.shape[2]
    cin = sum(p.shape[1] for p in parts)
    c1, classes = conv.out_channels, ctr_reg.out_channels
    cin_pad = _pad16(cin)
    if cin != conv.in_channels or c1 % 16 or M % 16 or classes > 16 or ctr_reg.in_channels != c1:
        return None
    if 17 * 4 * (max(cin_pad, c1) + c1) > 150 * 1024:
        return None
    key = _version_key([(conv, bn)], x0.device) + tuple(
        (t.data_ptr(), t._version) for t in (ctr_reg.weight, ctr_reg.bias) if t is not None)
    packed = getattr(mlp, "_sps_vote", None)
    if packed is None or packed.key != key:
        with torch.no_grad():
            packed = PackedTail()
            packed.key, packed.cin, packed.c1, packed.c2, packed.classes = key, cin_pad, c1, c1, classes
            w, b = _fold(conv, bn)
            wp = w.new_zeros(c1, cin_pad)
            wp[:, :cin] = w                      # padded input channels read zeros and weigh nothing
            packed.w1, packed.b1 = _pack_pw(wp, c1), b.contiguous()
            packed.w2, packed.b2 = _pack_pw(torch.eye(c1, dtype=torch.float32, device=w.device), c1), b.new_zeros(c1)
            w3 = ctr_reg.weight.detach().reshape(classes, c1).float()
            b3 = ctr_reg.bias.detach().float() if ctr_reg.bias is not None else w3.new_zeros(classes)
            packed.w3, packed.b3 = _pack_pw(w3, 16), _pad_bias(b3, 16)
        object.__setattr__(mlp, "_sps_vote", packed)
    pieces = list(parts)
    if cin_pad != cin:
        pieces.append(x0.new_zeros((B, cin_pad - cin, M)))
    x = torch.cat(pieces, dim=1) if len(pieces) > 1 else x0.contiguous()
    y1 = torch.empty((B, c1, M), dtype=torch.float32, device=x.device)
    y3 = torch.empty((B, M, classes), dtype=torch.float32, device=x.device)
    _lib.check(_L.sps_pointwise_mlp(B, M, cin_pad, c1, c1, classes, x.data_ptr(), packed.w1.data_ptr(), packed.b1.data_ptr(),
                                    packed.w2.data_ptr(), packed.b2.data_ptr(), packed.w3.data_ptr(), packed.b3.data_ptr(),
                                    y1.data_ptr(), 0, y3.data_ptr(), _lib.raw_stream(x.device)),
               "pointwise_mlp (vote)")
    return y3


# ---- DenseEdgeConv / FeatureExtraction (surface_feature.py:45-187) ------------------------------------------------------
class PackedEdgeConv:
    __slots__ = ("key", "w1", "b1", "w2", "b2", "w3", "b3", "relative", "form")


# True: evaluate a full convolution's first layer with merged weight blocks (a third fewer MFMAs, ~1e-6 relative off).
# Off by default: dynamic graphs feed the features back as query positions, where last-bit changes flip neighbours.
DEC_MERGED = False


def _dec_frag(w, cols):
    """w (12, C) -> (len(cols), 64): element [ks][16 q + i] = w[i][cols[ks][q]] (rows >= 12 and column -1 read as zero)."""
    g = w.shape[0]
    wp = torch.zeros((16, w.shape[1] + 1), dtype=torch.float32, device=w.device)
    wp[:g, :-1] = w
    col = torch.as_tensor(cols, dtype=torch.long, device=w.device)          # (KS, 4), -1 -> the zero column
    col = torch.where(col < 0, torch.full_like(col, w.shape[1]), col)
    return wp[:, col].permute(1, 2, 0).reshape(len(cols), 64).contiguous()  # (16, KS, 4) -> (KS, 4, 16)


def dense_edge_conv_supported(conv):
    return (conv.in_channels == 24 and conv.knn == 16 and conv.growth_rate == 12 and conv.num_fc_layers == 3
            and conv.aggr.oper == 'max' and isinstance(conv.layer_first.activation, nn.ReLU)
            and isinstance(conv.layers[0].activation, nn.ReLU) and isinstance(conv.layer_last.activation, nn.Identity)
            and all(l.linear.bias is not None for l in (conv.layer_first, conv.layers[0], conv.layer_last)))


def pack_dense_edge_conv(conv):
    """Weights of a DenseEdgeConv in the fragment order of csrc/dense_edge_conv.hip: input channel 6q + j of the 24 wide
    blocks sits in k-slot q of k-step j; the 12 activation channels of a previous layer sit as 4q + r (q < 3)."""
    lins = (conv.layer_first.linear, conv.layers[0].linear, conv.layer_last.linear)
    key = tuple((t.data_ptr(), t._version) for l in lins for t in (l.weight, l.bias)) + (DEC_MERGED,)
    packed = getattr(conv, "_sps_packed", None)
    if packed is not None and packed.key == key:
        return packed
    with torch.no_grad():
        wide = lambda base: [[base + 6 * q + j for q in range(4)] for j in range(6)]
        act = lambda base: [[base + 4 * q + r if q < 3 else -1 for q in range(4)] for r in range(4)]
        packed = PackedEdgeConv()
        packed.key, packed.relative = key, bool(conv.relative_feat_only)
        w1, w2, w3 = (l.weight.detach().float() for l in lins)
        if packed.relative:
            packed.w1, packed.form = _dec_frag(w1, wide(0)), 1
        elif DEC_MERGED:  # [x_i | x_j | x_j - x_i] -> [(W1a - W1c) x_i | (W1b + W1c) x_j]
            merged = torch.cat([w1[:, 0:24] - w1[:, 48:72], w1[:, 24:48] + w1[:, 48:72]], dim=1)
            packed.w1, packed.form = _dec_frag(merged, wide(0) + wide(24)), 2
        else:
            packed.w1, packed.form = _dec_frag(w1, wide(0) + wide(24) + wide(48)), 0
        packed.w2 = _dec_frag(w2, act(0) + wide(12))
        packed.w3 = _dec_frag(w3, act(0) + act(12) + wide(24))
        packed.b1, packed.b2, packed.b3 = (_pad_bias(l.bias.detach().float(), 16) for l in lins)
    object.__setattr__(conv, "_sps_packed", packed)
    return packed


def dense_edge_conv(conv, x, idx):
    """x (B, N, 24) fp32, idx (B, N, 16) int32 -> (B, N, 60); one launch (sps_dense_edge_conv)."""
    packed = pack_dense_edge_conv(conv)
    B, N, d = x.shape
    x = x.contiguous()
    out = torch.empty((B, N, conv.out_channels), dtype=torch.float32, device=x.device)
    _lib.check(_L.sps_dense_edge_conv(B, N, d, idx.shape[2], conv.growth_rate, packed.form, x.data_ptr(),
                                      idx.data_ptr(), packed.w1.data_ptr(), packed.b1.data_ptr(), packed.w2.data_ptr(),
                                      packed.b2.data_ptr(), packed.w3.data_ptr(), packed.b3.data_ptr(), out.data_ptr(),
                                      _lib.raw_stream(x.device)), "dense_edge_conv")
    return out


def linear_rows_supported(fc):
    return (fc.linear.out_features == 24 and fc.linear.in_features <= 64
            and isinstance(fc.activation, (nn.ReLU, nn.Identity)))


def linear_rows(fc, x):
    """FCLayer on point-major rows: x (..., cin) -> (..., 24); one launch (sps_linear_rows)."""
    lin = fc.linear
    x = x.contiguous()
    rows = x.numel() // x.shape[-1]
    out = torch.empty(x.shape[:-1] + (lin.out_features,), dtype=torch.float32, device=x.device)
    _lib.check(_L.sps_linear_rows(rows, lin.in_features, lin.out_features, x.data_ptr(), lin.weight.data_ptr(),
                                  0 if lin.bias is None else lin.bias.data_ptr(), int(isinstance(fc.activation, nn.ReLU)),
                                  out.data_ptr(), _lib.raw_stream(x.device)), "linear_rows")
    return out


def _dec_frag_t(mt):
    """mt (rows <= 32, k <= 12): a transposed weight block -> (4 * tiles, 64) fragments, element [4 t + r][16 q + i] =
    mt[16 t + i][4 q + r] (zero outside): the A operand of `mt @ dZ` with dZ in the accumulator layout."""
    rows, k = mt.shape
    tiles = (rows + 15) // 16
    pad = mt.new_zeros((16 * tiles, 16))
    pad[:rows, :k] = mt
    # [t, i, q, r] -> [t, r, q, i]
    return pad.view(tiles, 16, 4, 4).permute(0, 3, 2, 1).reshape(4 * tiles, 64).contiguous()


def _dec_bwd_maps(conv, device):
    """Index maps that turn the flat parameter vector [0, w1, w2, w3] into the forward and transposed fragments of
    csrc/dense_edge_conv_bwd.hip (built once per module by running the packing on matrices of indices): in training the
    weights change every step, and packing them with ~40 small torch ops cost more host time than the kernels."""
    maps = getattr(conv, "_sps_bwd_maps", None)
    if maps is not None and maps[0].device == device:
        return maps
    lins = (conv.layer_first.linear, conv.layers[0].linear, conv.layer_last.linear)
    shapes = [tuple(l.weight.shape) for l in lins]
    sizes = [a * b for a, b in shapes]
    offs = [1, 1 + sizes[0], 1 + sizes[0] + sizes[1]]   # index 0 is a zero
    i1, i2, i3 = (torch.arange(n, dtype=torch.float32, device=device).add_(o).view(sh) for n, o, sh in zip(sizes, offs, shapes))
    wide = lambda base: [[base + 6 * q + j for q in range(4)] for j in range(6)]
    act = lambda base: [[base + 4 * q + r if q < 3 else -1 for q in range(4)] for r in range(4)]
    rel = bool(conv.relative_feat_only)
    f1 = _dec_frag(i1, wide(0) if rel else wide(0) + wide(24) + wide(48))
    perm_f = torch.cat([f1, _dec_frag(i2, act(0) + wide(12)), _dec_frag(i3, act(0) + act(12) + wide(24))], dim=0)
    zero = lambda like: torch.zeros_like(like)
    blocks = [(i3[:, 0:12].t(), 1.0, None, 0.0), (i3[:, 12:24].t(), 1.0, None, 0.0), (i2[:, 0:12].t(), 1.0, None, 0.0),
              (i3[:, 24:48].t(), 1.0, None, 0.0), (i2[:, 12:36].t(), 1.0, None, 0.0)]
    if rel:
        blocks += [(i1[:, 0:24].t(), -1.0, None, 0.0), (i1[:, 0:24].t(), 1.0, None, 0.0)]
    else:
        blocks += [(i1[:, 0:24].t(), 1.0, i1[:, 48:72].t(), -1.0), (i1[:, 24:48].t(), 1.0, i1[:, 48:72].t(), 1.0)]
    pa = torch.cat([_dec_frag_t(a.contiguous()) for a, _, _, _ in blocks], dim=0)
    pb = torch.cat([_dec_frag_t((b if b is not None else zero(a)).contiguous()) for a, _, b, _ in blocks], dim=0)
    sa = torch.cat([torch.full_like(_dec_frag_t(a.contiguous()), s) for a, s, _, _ in blocks], dim=0)
    sb = torch.cat([torch.full_like(_dec_frag_t(a.contiguous()), s) for a, _, _, s in blocks], dim=0)
    maps = (perm_f.round().long().reshape(-1), pa.round().long().reshape(-1), sa.reshape(-1), pb.round().long().reshape(-1),
            sb.reshape(-1))
    object.__setattr__(conv, "_sps_bwd_maps", maps)
    return maps


def pack_dense_edge_conv_bwd(conv):
    """Fragments of csrc/dense_edge_conv_bwd.hip: (w_fwd, w_transposed, b1, b2, b3); the forward part is the reference
    form of the first layer ([x_i | x_j | x_j - x_i]), whatever DEC_MERGED says."""
    lins = (conv.layer_first.linear, conv.layers[0].linear, conv.layer_last.linear)
    dev = lins[0].weight.device
    perm_f, pa, sa, pb, sb = _dec_bwd_maps(conv, dev)
    with torch.no_grad():
        flat = torch.cat([lins[0].weight.new_zeros(1)] + [l.weight.detach().float().reshape(-1) for l in lins])
        wf = flat[perm_f].view(-1, 64)
        wt = (flat[pa] * sa + flat[pb] * sb).view(44, 64)
        bias = flat.new_zeros(48)
        for k, l in enumerate(lins):
            bias[16 * k:16 * k + l.bias.numel()] = l.bias.detach()
    return wf, wt, bias[0:16], bias[16:32], bias[32:48]


def dense_edge_conv_backward(conv, x, idx, grad_out, packed):
    """-> (dx (B, N, 24), dW1, db1, dW2, db2, dW3, db3) of dense_edge_conv(conv, x, idx)."""
    from . import pointnet2_batch_cuda as _ext
    wf, wt, b1, b2, b3 = packed
    B, N, d = x.shape
    rel = bool(conv.relative_feat_only)
    tiles = 9 if rel else 13
    dev = x.device
    dxc = torch.empty((B, N, d), dtype=torch.float32, device=dev)
    dxn = torch.empty((B, d, N, 16), dtype=torch.float32, device=dev)
    blocks = int(_L.sps_dense_edge_conv_bwd_blocks())
    partial = torch.empty((blocks, tiles * 256), dtype=torch.float32, device=dev)
    gt = torch.empty((tiles, 16, 16), dtype=torch.float32, device=dev)
    _lib.check(_L.sps_dense_edge_conv_bwd(B, N, d, 16, conv.growth_rate, int(rel), x.data_ptr(), idx.data_ptr(),
                                          grad_out.data_ptr(), wf.data_ptr(), wt.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                                          b3.data_ptr(), dxc.data_ptr(), dxn.data_ptr(), partial.data_ptr(), gt.data_ptr(),
                                          _lib.raw_stream(dev)), "dense_edge_conv_bwd")
    scat = torch.zeros((B, d, N), dtype=torch.float32, device=dev)
    _ext.group_points_grad_wrapper(B, d, N, N, 16, dxn, idx, scat)
    dx = dxc + scat.transpose(1, 2)
    g = 12
    dw3 = torch.cat([gt[0, :g, :12], gt[1, :g, :12], gt[2, :g, :16], gt[3, :g, :8]], dim=1)
    dw2 = torch.cat([gt[4, :g, :12], gt[5, :g, :16], gt[6, :g, :8]], dim=1)
    if rel:
        dw1 = torch.cat([gt[7, :g, :16], gt[8, :g, :8]], dim=1)
    else:
        dw1 = torch.cat([gt[7, :g, :16], gt[8, :g, :8], gt[9, :g, :16], gt[10, :g, :8], gt[11, :g, :16], gt[12, :g, :8]], dim=1)
    return dx, dw1, gt[8, :g, 8].clone(), dw2, gt[6, :g, 8].clone(), dw3, gt[3, :g, 8].clone()


class DenseEdgeConvTrain(torch.autograd.Function):
    """DenseEdgeConv.forward with gradients: fused forward kernel, fused recompute-and-backpropagate kernel."""

    @staticmethod
    def forward(ctx, x, idx, w1, b1, w2, b2, w3, b3, conv):
        x = x.contiguous()
        packed = pack_dense_edge_conv_bwd(conv)
        wf, _, pb1, pb2, pb3 = packed
        ks1 = 6 if conv.relative_feat_only else 18
        B, N, d = x.shape
        out = torch.empty((B, N, conv.out_channels), dtype=torch.float32, device=x.device)
        _lib.check(_L.sps_dense_edge_conv(B, N, d, idx.shape[2], conv.growth_rate, 1 if conv.relative_feat_only else 0,
                                          x.data_ptr(), idx.data_ptr(), wf[:ks1].data_ptr(), pb1.data_ptr(),
                                          wf[ks1:ks1 + 10].data_ptr(), pb2.data_ptr(), wf[ks1 + 10:].data_ptr(), pb3.data_ptr(),
                                          out.data_ptr(), _lib.raw_stream(x.device)), "dense_edge_conv")
        ctx.save_for_backward(x, idx)
        ctx.packed, ctx.conv = packed, conv
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, idx = ctx.saved_tensors
        dx, dw1, db1, dw2, db2, dw3, db3 = dense_edge_conv_backward(ctx.conv, x, idx, grad_out.contiguous(), ctx.packed)
        return dx, None, dw1, db1, dw2, db2, dw3, db3, None


class LinearRowsTrain(torch.autograd.Function):
    """FCLayer (Linear + optional ReLU) on point-major rows with gradients: sps_linear_rows / sps_linear_rows_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = x.contiguous()
        rows, cin = x.numel() // x.shape[-1], x.shape[-1]
        out = torch.empty(x.shape[:-1] + (weight.shape[0],), dtype=torch.float32, device=x.device)
        _lib.check(_L.sps_linear_rows(rows, cin, weight.shape[0], x.data_ptr(), weight.data_ptr(),
                                      0 if bias is None else bias.data_ptr(), int(relu), out.data_ptr(),
                                      _lib.raw_stream(x.device)), "linear_rows")
        ctx.save_for_backward(x, out, weight)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, y, weight = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = weight.shape
        rows = x.numel() // cin
        dx = torch.empty_like(x)
        blocks = int(_L.sps_linear_rows_bwd_blocks())
        partial = torch.empty((blocks, cout * cin + cout), dtype=torch.float32, device=x.device)
        gwb = torch.empty((cout * cin + cout,), dtype=torch.float32, device=x.device)
        _lib.check(_L.sps_linear_rows_bwd(rows, cin, cout, x.data_ptr(), y.data_ptr(), dy.data_ptr(), weight.data_ptr(),
                                          int(ctx.relu), dx.data_ptr(), partial.data_ptr(), gwb.data_ptr(),
                                          _lib.raw_stream(x.device)), "linear_rows_bwd")
        return dx, gwb[:cout * cin].view(cout, cin), (gwb[cout * cin:] if ctx.has_bias else None), None
