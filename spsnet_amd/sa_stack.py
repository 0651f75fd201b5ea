"""IA-SSD / SPSNet set-abstraction stack driver (what bench.py times and the parity tests check).

Builds the SA layers exactly as the reference's backbone does from its YAML
(pcdet/models/backbones_3d/IASSD_backbone.py:30-84, config
tools/cfgs/kitti_models/IA-SSD.yaml:35-55 / SPSNet.yaml:39-69) and drives them like
IASSD_backbone.py:128-134: layer k consumes layer k-1's (xyz, features, cls prediction).
Only the SA layers are built -- heads, losses and the voting branch stay out of scope.
"""
import copy
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib, streams

# the three down-sampling SA layers of BASELINE.json's metric (16 384 -> 4 096 -> 1 024 -> 512)
IASSD_KITTI = dict(
    npoint_list=[[4096], [1024], [512]],
    sample_range_list=[[-1], [-1], [-1]],
    sample_method_list=[['D-FPS'], ['D-FPS'], ['ctr_aware']],
    radius_list=[[0.2, 0.8], [0.8, 1.6], [1.6, 4.8]],
    nsample_list=[[16, 32], [16, 32], [16, 32]],
    mlps=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 128, 256], [128, 256, 256]]],
    aggregation_mlps=[[64], [128], [256]],
    confidence_mlps=[[], [128], [256]],
    dilated_group=[False, False, False],
)


def scaled_config(base=IASSD_KITTI, npoints=None, nsamples=None, sample_methods=None):
    cfg = copy.deepcopy(base)
    if npoints is not None:
        cfg['npoint_list'] = [[p] for p in npoints]
    if nsamples is not None:
        cfg['nsample_list'] = [list(ns) for ns in nsamples]
    if sample_methods is not None:
        cfg['sample_method_list'] = [[m] for m in sample_methods]
    return cfg


def build_sa_layers(modules_pkg, cfg=IASSD_KITTI, input_channels=4, num_class=3, seed=0) -> nn.ModuleList:
    """Instantiate the SA layers with `modules_pkg.PointnetSAModuleMSG_WithSampling` (the build's
    pointnet2_modules or, for golden generation, the reference's) under a fixed seed, eval mode,
    with non-trivial deterministic BatchNorm running statistics."""
    torch.manual_seed(seed)
    layers = nn.ModuleList()
    channel_in = input_channels - 3
    for k in range(len(cfg['npoint_list'])):
        mlps = [[channel_in] + list(m) for m in cfg['mlps'][k]]
        channel_out = sum(m[-1] for m in mlps)
        agg = list(cfg['aggregation_mlps'][k]) or None
        if agg:
            channel_out = agg[-1]
        conf = list(cfg['confidence_mlps'][k]) or None
        layers.append(modules_pkg.PointnetSAModuleMSG_WithSampling(
            npoint_list=cfg['npoint_list'][k], sample_range_list=cfg['sample_range_list'][k],
            sample_type_list=cfg['sample_method_list'][k], radii=cfg['radius_list'][k],
            nsamples=cfg['nsample_list'][k], mlps=mlps, use_xyz=True,
            dilated_group=cfg['dilated_group'][k], aggregation_mlp=agg, confidence_mlp=conf,
            num_class=num_class))
        channel_in = channel_out
    gen = torch.Generator().manual_seed(seed + 1)
    for mod in layers.modules():
        if isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d)):
            with torch.no_grad():
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=gen) + 0.5)
                mod.weight.copy_(torch.rand(mod.num_features, generator=gen) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_features, generator=gen) * 0.1)
    return layers.eval()


HYBRID = False          # see _streamed_first_layer
FUSE_GATHER = os.environ.get("SPS_FUSE_GATHER", "1") != "0"   # streamed chunks: the ball query gathers its own centroids
# training passes: layer 0's ball queries consume the publishing FPS while it samples (_streamed_first_layer_queries)
STREAM_TRAINING_QUERIES = os.environ.get("SPS_STREAM_TRAINING_QUERIES", "1") != "0"
_SIDE_STREAMS = {}
_FENCES = {}          # (device index, main stream handle) -> CuFence


class CuFence:
    """Compute-unit partition for the passes issued on one main stream: the serial FPS chain of a pass gets compute units of
    its own, every helper stream of the pass is kept off them.  FPS occupies one CU per scene for most of a pass and
    whatever shares a CU with one of its workgroups slows that chain (two waves on a SIMD cost each other ~25 %); with two
    passes in flight the second pass's whole-chip kernels otherwise land exactly there.

    Mask numbering as measured on MI355X (tools/cumask_probe.py, profiles/round2/cumask_probe_gfx950.txt): bit i of a
    hipExtStreamCreateWithCUMask mask is CU i // 8 of XCD i % 8, and an XCD WITHOUT any bit set is not restricted at all --
    so every mask here carries bits for all eight XCDs.  Slot k of `slots` concurrent passes owns CU indices
    [k c, (k + 1) c) of every XCD, c = ceil(scenes / 8) (workgroups are dealt round-robin over the XCDs: one FPS
    workgroup per owned CU); helper streams get CU indices >= slots * c."""

    XCDS, CUS_PER_XCD = 8, 32

    def __init__(self, device, slot=0, slots=1, scenes=8):
        from . import _lib
        self.device = device
        per = max(1, -(-scenes // self.XCDS))
        if slots * per >= self.CUS_PER_XCD:
            raise ValueError("no compute units left for the helper streams")
        self.fps_bits = [cu * self.XCDS + x for cu in range(slot * per, (slot + 1) * per) for x in range(self.XCDS)]
        self.rest_bits = [cu * self.XCDS + x for cu in range(slots * per, self.CUS_PER_XCD) for x in range(self.XCDS)]
        self._lib, self._handles = _lib, []

    def stream(self, bits):
        import ctypes
        words = self.XCDS * self.CUS_PER_XCD // 32
        mask = (ctypes.c_uint * words)()
        for b in bits:
            mask[b // 32] |= 1 << (b % 32)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.load().sps_stream_create_cu_mask(words, ctypes.cast(mask, ctypes.c_void_p),
                                                                       ctypes.byref(handle)), "stream_create_cu_mask")
        self._handles.append(handle)      # owned for the life of the process (a handful of streams)
        return torch.cuda.ExternalStream(handle.value, device=self.device)

    def new_stream(self, tag):
        return self.stream(self.fps_bits if tag == "fps" else self.rest_bits)


def enable_cu_fence(device, main_stream=None, slot=0, slots=1, scenes=8):
    """Fence the FPS chain of every streamed pass issued on `main_stream` (default: the device's current stream) onto its
    own compute units; -> the CuFence (use .stream(.rest_bits) to make a main stream for slot k of a pipelined server)."""
    device = torch.device(device)
    main = torch.cuda.current_stream(device) if main_stream is None else main_stream
    fence = CuFence(device, slot, slots, scenes)
    _FENCES[(device.index, main.cuda_stream)] = fence
    for key in [k for k in _SIDE_STREAMS if k[1] == device.index and k[2] == main.cuda_stream]:
        del _SIDE_STREAMS[key]
    return fence


def disable_cu_fence(device, main_stream=None):
    device = torch.device(device)
    main = torch.cuda.current_stream(device) if main_stream is None else main_stream
    _FENCES.pop((device.index, main.cuda_stream), None)
    for key in [k for k in _SIDE_STREAMS if k[1] == device.index and k[2] == main.cuda_stream]:
        del _SIDE_STREAMS[key]


def _drop_cached_streams(device_index, root_handle, helper_handles):
    """streams.forget(): the roles cached here for that pass (asked for from its main stream or from one of its helpers)"""
    gone = set(helper_handles) | {root_handle}
    for key in [k for k in _SIDE_STREAMS if k[1] == device_index and k[2] in gone]:
        del _SIDE_STREAMS[key]


streams.on_forget(_drop_cached_streams)


def _helper_stream(device, tag="side"):
    """One helper stream per (device, current stream, role): concurrent passes on different streams do not share it.
    With a CuFence registered for the current stream the helper streams are CU-masked; tag "fps" = the stream the FPS
    producer runs on (None without a fence: the producer then stays on the main stream)."""
    main = torch.cuda.current_stream(device)
    fence = _FENCES.get((device.index, main.cuda_stream))
    if tag == "fps" and fence is None:
        return None
    key = (device.type, device.index, main.cuda_stream, tag)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (fence.new_stream(tag) if fence is not None else
                              streams.helper(device, main, tag, exclusive=(tag == "producer")))
    return _SIDE_STREAMS[key]


def _side_stream(device):
    return _helper_stream(device, "side")


def _prefetch_dfps(next_layer, new_xyz, fps_ordered=False):
    """Start the NEXT layer's D-FPS on a side stream as soon as this layer's centroids exist: it needs only
    their coordinates, so it runs beside this layer's ball queries and grouped MLPs (FPS occupies one CU per
    scene; the rest of the chip is free).  The next layer's sampler waits on the event."""
    from . import pointnet2_utils
    main = torch.cuda.current_stream(new_xyz.device)
    side = _side_stream(new_xyz.device)
    ready = torch.cuda.Event()
    ready.record(main)
    new_xyz.record_stream(side)
    with torch.cuda.stream(side), torch.no_grad():
        side.wait_event(ready)
        if fps_ordered:
            # new_xyz is a D-FPS pick sequence: its own D-FPS is the identity prefix up to distance ties, which
            # sps_fps_ordered_prefix verifies in parallel (and recomputes where it fails) -- same result, ~20x faster
            from . import pointnet2_batch_cuda as _ext
            idx = _ext.fps_ordered_prefix(new_xyz, next_layer.npoint_list[0])
        else:
            idx = pointnet2_utils.furthest_point_sample(new_xyz, next_layer.npoint_list[0])
        done = torch.cuda.Event()
        done.record(side)
    next_layer._presampled = (idx, done, new_xyz)


def _can_prefetch(layer, nxt):
    types = getattr(nxt, "sample_type_list", None)
    if not types or len(types) != 1 or nxt.sample_range_list[0] != -1:
        return False
    t = types[0]
    is_dfps = ('D-FPS' in t or 'DFS' in t) and not ('cls' in t or 'ctr' in t or 'ss' in t)
    return is_dfps and sum(p for p in layer.npoint_list if p > 0) > nxt.npoint_list[0] > 0


# centroid chunks of the streamed first layer, as cumulative fractions of M in 1/16ths: big chunks while FPS still has a
# long way to go, small ones at the end (what is left to do after the last pick is the last chunk's work; a chunk
# costs ~70 us however small it is -- one wave scans the whole cloud per centroid -- so finer tails only add launches).
# (3/16 steps up to 12/16 since round 3: they leave the consumer stream the idle windows the next layer's early stages need,
#  see EARLY_POOL_AT_16; the pass without those stages takes the same time with either chunking.)
def _parse_chunk_ends(text):
    """SPS_CHUNK_ENDS: strictly ascending sixteenths ending in 16 -- a list that stops short never reaches the last chunk (no
    repair of timed-out waits, no `verify.finish`, centroids beyond its last entry never computed)."""
    try:
        ends = tuple(int(v) for v in text.split(","))
    except ValueError:
        raise ValueError(f"SPS_CHUNK_ENDS={text!r}: expected comma-separated integers (sixteenths of the layer's centroids)")
    if not ends or ends[-1] != 16 or ends[0] < 1 or any(b <= a for a, b in zip(ends, ends[1:])):
        raise ValueError(f"SPS_CHUNK_ENDS={text!r}: expected strictly ascending values in 1..16 whose last entry is 16")
    return ends


def _parse_early_pool_at(text, ends):
    """SPS_EARLY_POOL_AT: the chunk ends behind which the next layer's early stages fire -- each must BE a chunk end (a stage
    only fires where a chunk ends) and lie before the last one (behind the last pick the layer itself runs)."""
    try:
        at = tuple(int(v) for v in text.split(",") if v.strip())
    except ValueError:
        raise ValueError(f"SPS_EARLY_POOL_AT={text!r}: expected comma-separated integers (sixteenths), or nothing")
    if any(b <= a for a, b in zip(at, at[1:])) or any(e not in ends[:-1] for e in at):
        raise ValueError(f"SPS_EARLY_POOL_AT={text!r}: expected ascending values out of the chunk ends {ends[:-1]} (SPS_CHUNK_ENDS)")
    return at


_CHUNK_ENDS_16 = _parse_chunk_ends(os.environ.get("SPS_CHUNK_ENDS", "3,6,9,12,14,15,16"))
# the chunk ends (in 1/16ths of M) behind which the NEXT layer starts / continues on the picks that exist (begin_early_pool).
# The producer needs 0.33 ms per 3/16 of its picks, a 3/16 chunk's own work takes ~0.11 ms of that: the stages' queries and
# grouped MLPs (~0.12 ms each) fit the idle time behind the chunks that end at 6, 9 and 12 sixteenths.  Measured
# (tools/tail_events.py, strict fp32, 8 x 16 384): no stage 2.394 ms per pass, stages at (6, 9, 12) 2.305; with the old
# chunking (4, 8, 12, 14, 15, 16) and stages at (8, 12) 2.355 -- the second stage pushed the last chunks back by 0.05 ms --,
# and a stage behind the 14/16 chunk costs more than it saves (+0.08 ms).
EARLY_POOL_AT_16 = _parse_early_pool_at(os.environ.get("SPS_EARLY_POOL_AT", "6,9,12"), _CHUNK_ENDS_16)
# the next layer's last stage gathers its verified centroids inside its ball query instead of a launch in front of it
LATE_GATHER = os.environ.get("SPS_LATE_GATHER", "1") != "0"
# the second pass of the next layer's identity-prefix check chunk by chunk beside the producer instead of whole behind it
EARLY_PREFIX_CHECK = os.environ.get("SPS_EARLY_PREFIX_CHECK", "1") != "0"
_TIMEOUT_FLAGS = []   # device flags of recent streamed passes (diagnostics only: see check_timeouts)


def check_timeouts():
    """True if a device-side progress wait of a recent streamed pass gave up (synchronises; tests / bench / debugging).
    Not an error: such a pass repaired itself (the predicated redo in _streamed_first_layer) and its results are correct;
    it only took longer."""
    bad = any(bool(f.any().item()) for f in _TIMEOUT_FLAGS)
    _TIMEOUT_FLAGS.clear()
    return bad


def _streamed_first_layer(layer, nxt, xyz, features, stds=None, after_producer=None):
    """Layer 0 with its grouping/MLP consuming the D-FPS output WHILE the FPS kernel is still running.

    FPS is a serial chain on one CU per scene (~2.9 ms for 8 x 16384 -> 4096); the ball queries and grouped MLPs of
    the centroids it has already produced need nothing else, so they run chunk by chunk on a second stream, each
    chunk gated by a bounded device-side wait on the kernel's published progress counter (csrc: sps_fps_publish /
    sps_wait_progress, the write-through hand-off of the CDNA guide).  Same kernels, same results; only the
    schedule differs.  Returns None when the layer does not qualify (caller falls back to layer.forward).
    after_producer: called on the host right after the FPS kernel has been enqueued -- the place for independent work
    that should run beside it (PAGNet's surface features): enqueued earlier it delays the FPS launch by its own host
    time, enqueued after this function it starts ~1.7 ms late (the host time of the chunk pipeline below)."""
    from . import fused as _fused
    from . import pointnet2_batch_cuda as _ext
    from . import pointnet2_utils
    B, N, _ = xyz.shape
    M = layer.npoint_list[0]
    if not (_is_plain_dfps(layer, N) and _ext.fps_can_publish(B, N) and M % (64 * 16) == 0 and len(layer.groupers) == 2):
        return None
    if layer.training or features is None or not xyz.is_contiguous() or not features.is_contiguous():
        return None
    dev = xyz.device
    new_xyz = torch.empty((B, M, 3), dtype=torch.float32, device=dev)
    plan = layer._fused_plan(xyz, new_xyz, features)
    if not plan:
        return None
    ga, gb = layer.groupers
    main = torch.cuda.current_stream(dev)
    fps_stream = _helper_stream(dev, "fps")              # None unless a CuFence is registered for `main`
    if fps_stream is not None and N > 16384:
        # the clustered large-scene producer spins on its sibling workgroups: all B K of them must be resident at once, which a
        # stream confined to a few compute units does not promise -- large scenes run unstreamed under a CuFence
        return None
    # Who runs where (measured, tools/hop_cost.py: a kernel boundary costs ~4 us, a cross-stream dependency that is actually
    # waited for ~12 us more, one that was satisfied long before nothing):
    #   default   every chunk on the caller's stream behind bounded waits (nothing downstream has to hop streams); the producer on
    #             a helper stream -- with a CuFence registered, on a stream of its own compute units.  One live dependency: the
    #             producer's start.                                                      2.264 ms per pass at the bench shape
    #   HYBRID    the producer on the caller's stream, the chunks that overlap it on a helper stream, the last chunk back on
    #             the caller's stream in order behind the producer (no wait kernel, no live dependency).  Measured SLOWER,
    #             2.296 ms: the last chunk then starts at the producer kernel's end instead of at its last published sample.
    masked = fps_stream is not None                      # a CuFence stream: a few compute units only
    if fps_stream is None and not HYBRID:
        fps_stream = _helper_stream(dev, "producer")
    fenced = fps_stream is not None
    side = main if fenced else _helper_stream(dev, "chunks")

    idx = torch.empty((B, M), dtype=torch.int32, device=dev)
    temp = None                                          # the producer starts from 1e10 by itself and hands nothing back
    zeros = _zeroed_ints(dev, 2 * B)                     # zero BEFORE the consumer is released; no fill launch per pass
    progress, timed_out = zeros[:B], zeros[B:]           # timed_out: one flag per scene, set together
    idx_a = torch.empty((B, M, ga.nsample), dtype=torch.int32, device=dev)
    idx_b = torch.empty((B, M, gb.nsample), dtype=torch.int32, device=dev)
    out = (torch.zeros if max(ga.nsample, gb.nsample) > 32 else torch.empty)(
        (B, sum(p.c3_real for p in plan), M), dtype=torch.float32, device=dev)
    _TIMEOUT_FLAGS.append(timed_out)
    del _TIMEOUT_FLAGS[:-64]

    # the aggregation / confidence tail runs chunk by chunk behind the grouped MLPs (its 16-point tiles are independent),
    # so that only the last chunk's share of it is left when FPS ends.  Its runner is built HERE, before the producer is
    # launched: on the first call of a module it folds and packs the weights with torch ops on the main stream, and the
    # consumer stream below is ordered after `start` only -- packed behind the FPS kernel, the weights reached the early
    # chunks too late (the first pass of freshly built modules returned wrong features for all chunks but the last).
    half = features.dtype == torch.float16
    tail = _fused.tail_runner(layer.aggregation_layer, getattr(layer, "confidence_layers", None), out, half)
    _fused._overflow_flag(dev)                           # (created on first use: also before `start`)
    start = torch.cuda.Event()
    start.record(main)
    if not fenced:
        _ext.fps_publish(xyz, temp, idx, progress)      # producer, on the caller's stream
    else:
        fps_stream.wait_event(start)
        with torch.cuda.stream(fps_stream):
            # (the workspace of the sorting pre-pass / of the large-scene kernel; no pre-pass on a CU-masked stream: its
            #  workgroups spin on each other and must all be resident)
            work = _ext.fps_publish(xyz, temp, idx, progress, presort=not masked)
        for t in (xyz, idx, progress) + (() if work is None else (work,)):
            t.record_stream(fps_stream)
    if after_producer is not None:
        after_producer()
    if side is not main:
        for t in (xyz, features, idx, progress, timed_out, new_xyz, idx_a, idx_b, out) + (tail.tensors() if tail else ()):
            t.record_stream(side)
    ends = [M * e // 16 for e in _CHUNK_ENDS_16]
    # the next layer's D-FPS over these centroids is the verified identity prefix (fps_verify.hip): its first pass
    # needs only the first npoint centroids and runs as soon as they exist, its second when the last one does
    verify = None
    if nxt is not None and _can_prefetch(layer, nxt) and nxt.npoint_list[0] <= _ext.ORDERED_PREFIX_MAX:
        with torch.cuda.stream(side):
            verify = _ext.OrderedPrefix(new_xyz, nxt.npoint_list[0])
        for t in verify.tensors():
            t.record_stream(main)

    def consume(j0, j1, wait):
        """gather -> ball query -> grouped MLPs -> aggregation for the centroids [j0, j1) of every scene, on the current stream"""
        nonlocal xyz_ready, verify, verified_inline, guess
        chunk = j1 - j0
        if wait:
            _ext.wait_progress(progress, j1, timed_out, patient=(j1 == M))
        # Correct or redo, never invalid: the waits are bounded (a producer that stalls must not hang the device); one that
        # gave up let its consumers run on samples that had not been written.  The LAST chunk -- behind the patient wait, or
        # behind the producer itself -- therefore covers the whole layer when a flag is up (full_range_if: the kernels read
        # it and widen their range; no extra launch, nothing changes when every wait was served).
        repair = timed_out if (j1 == M and self_repair) else None
        if fuse_gather:
            # the ball query gathers its centroids itself (and writes new_xyz): one launch less per chunk -- and in the last
            # chunk the widened range re-gathers every centroid, so the predicated repair gather goes too
            _ext.ball_query_full2_range(ga.radius, gb.radius, xyz, new_xyz, idx_a, idx_b, j0, chunk, full_range_if=repair,
                                        gather_idx=idx)
        else:
            _ext.gather_xyz_range(xyz, idx, new_xyz, j0, chunk)
            if j1 == M:
                # every sample exists now: centroids that an earlier, timed-out wait let through are gathered again before
                # anybody else looks at them (a launch that does nothing otherwise)
                _ext.gather_xyz_range(xyz, idx, new_xyz, 0, M, run_if=timed_out)
            _ext.ball_query_full2_range(ga.radius, gb.radius, xyz, new_xyz, idx_a, idx_b, j0, chunk, full_range_if=repair)
        if verify is not None and j0 < verify.npoint <= j1:
            verify.begin()
        if j1 == M:
            if verify is not None and verify.checked > 0:
                # the centroids of the earlier chunks went through the second pass as they arrived (below): what is left is
                # this chunk's share, ~5 us -- in line, so that the next layer's sampling costs no stream hop at all
                picks = verify.finish(force_redo=timed_out)
                if LATE_GATHER and guess is not None and getattr(nxt, "_prepooled", None) is not None \
                        and getattr(nxt, "_on_new_xyz", None) is None:
                    # the next layer's centroids: the guess its early stages ran on, corrected IN PLACE by its last ball query,
                    # which gathers xyz[picks] itself (sps_ball_query_full2_points_gather) -- no gather launch on the chain
                    nxt._presampled = (picks, None, new_xyz, guess, True)
                else:
                    nxt._presampled = (picks, None, new_xyz)
                verify, verified_inline = None, True
            else:
                xyz_ready = torch.cuda.Event()
                xyz_ready.record(torch.cuda.current_stream(dev))
        off = 0
        for ix, packed in zip((idx_a, idx_b), plan):
            _fused.group_mlp_pool(xyz, new_xyz, features, ix, packed, out, off, j0, chunk, full_range_if=repair)
            off += packed.c3_real
        if tail is not None:
            tail.run(j0, chunk, full_range_if=repair)
        if j1 in early_at and verify is not None and verify.begun and tail is not None and side is main:
            # The NEXT layer starts here: its centroids are the first picks of this layer's D-FPS (the identity prefix that
            # `verify` confirms -- or flags -- behind the last pick), its cloud are this layer's centroids, of which the first
            # j1 exist with their features.  Its ball query over those, and the grouped MLP of the columns they give, run now,
            # beside the producer; behind the last pick only the columns of the remaining M - j1 points are left
            # (pointnet2_modules.begin_early_pool; exact -- max-pooling does not care which launch computed a column).
            # Correct or redo: should a wait have given up (timed_out) or the prefix guess fail (verify.flags), the late
            # launches recompute the layer from scratch.
            feats_next, _ = tail.result()
            if guess is None:
                guess = new_xyz[:, :nxt.npoint_list[0]].contiguous()
            nxt.begin_early_pool(new_xyz, guess, feats_next, j1, timed_out, verify.flags)
        if verify is not None and j1 < M and EARLY_PREFIX_CHECK and N <= 16384:
            # second pass of the next layer's identity-prefix check for the centroids that exist by now: the stream would
            # otherwise idle in the next chunk's wait (a chunk whose wait gave up checks garbage -- finish() is then told to
            # recompute every scene: force_redo).  Not beside the clustered large-scene producer: its workgroups wait on each
            # other every round, and pieces that stage 4 096 centres per workgroup on their compute units cost it 0.5 ms of
            # 7.8 (config 5, measured) -- there the whole pass runs behind the last pick on a third stream, as before.
            verify.check_upto(j1)

    xyz_ready, verified_inline = None, False
    early_at = tuple(M * e // 16 for e in EARLY_POOL_AT_16) if (nxt is not None and hasattr(nxt, "begin_early_pool") and N <= 16384) else ()
    guess = None
    # units of one centroid meet through an atomic max when a ball has more than 32 samples: a repair then needs `out`
    # zeroed again, which takes the separate predicated launches of _redo_layer
    self_repair = max(ga.nsample, gb.nsample) <= 32
    bounds = list(zip([0] + ends[:-1], ends))
    fuse_gather = FUSE_GATHER and self_repair and N >= 256 and all((j1 - j0) % 4 == 0 and B * (j1 - j0) <= 8192 for j0, j1 in bounds)
    with torch.cuda.stream(side):                        # the chunks that run beside the producer
        if side is not main:
            side.wait_event(start)
        for j0, j1 in (bounds if fenced else bounds[:-1]):
            consume(j0, j1, wait=True)
        if side is not main:
            early = torch.cuda.Event()
            early.record(side)
    if not fenced:                                       # the last chunk: behind the producer, on the caller's stream
        main.wait_event(early)                           # (satisfied long before the producer ends: costs nothing)
        consume(bounds[-1][0], M, wait=False)
    if not self_repair:   # the whole layer once more with run_if = timed_out: launches that do nothing when every wait was served
        _redo_layer(layer, plan, tail, xyz, new_xyz, features, idx_a, idx_b, out, timed_out)
    if verify is not None:
        third = _side_stream(dev)                        # beside the last chunk's ball query, not behind it
        with torch.cuda.stream(third):
            third.wait_event(xyz_ready)
            nidx = verify.finish(force_redo=timed_out)
            vdone = torch.cuda.Event()
            vdone.record(third)
        for t in verify.tensors() + (new_xyz,):
            t.record_stream(third)
        nxt._presampled = (nidx, vdone, new_xyz)   # (derived from the repaired centroids; flagged scenes are recomputed)
    elif nxt is not None and _can_prefetch(layer, nxt) and not verified_inline:
        _prefetch_dfps(nxt, new_xyz, True)
    if fenced and torch.cuda.is_current_stream_capturing():
        # a stream capture must end with every forked stream joined back; the consumers above are ordered behind the producer
        # by its published progress, not by an event (in eager mode nothing needs to wait for the kernel's exit)
        main.wait_stream(fps_stream)
    new_features, cls = tail.result() if tail is not None else layer._tail(out, half)
    if stds is not None:  # the layer's sampler thins the stability scores with its picks (reference :307-310)
        stds = pointnet2_utils.gather_operation(stds.view(B, 1, -1).contiguous(), idx).squeeze()
    return new_xyz, new_features, cls, idx, stds


def _streamed_first_layer_queries(layer, xyz):
    """TRAINING counterpart of _streamed_first_layer: the layer runs op by op (BatchNorm needs the statistics of ALL its grouped
    points before anything behind the first convolution can start), but its ball queries need the picks only -- they consume
    the publishing FPS chunk by chunk while it samples (1.8 ms on one CU per scene at 8 x 16384 -> 4096, with the other 248
    CUs idle), so that what is left behind the last pick is the last chunk's query.  Hands the picks, the gathered centroids
    and both scales' neighbour rows to the layer (`_presampled` / `_preball`); same kernels as the inference schedule, same
    rows as the layer's own ball queries, correct or redo (the last chunk widens to the whole layer when a bounded wait gave
    up).  Returns False when the layer does not qualify."""
    from . import pointnet2_batch_cuda as _ext
    from . import pointnet2_utils
    B, N, _ = xyz.shape
    M = layer.npoint_list[0] if getattr(layer, "npoint_list", None) else 0
    if not (xyz.is_cuda and xyz.dtype == torch.float32 and xyz.is_contiguous() and not xyz.requires_grad
            and _is_plain_dfps(layer, N) and _ext.fps_can_publish(B, N) and M % (64 * 16) == 0
            and len(layer.groupers) == 2 and all(type(g) is pointnet2_utils.QueryAndGroup for g in layer.groupers)):
        return False
    ga, gb = layer.groupers
    if max(ga.nsample, gb.nsample) > 32:
        return False
    dev = xyz.device
    bounds = list(zip([0] + [M * e // 16 for e in _CHUNK_ENDS_16[:-1]], [M * e // 16 for e in _CHUNK_ENDS_16]))
    if not (N >= 256 and all((j1 - j0) % 4 == 0 and B * (j1 - j0) <= 8192 for j0, j1 in bounds)):
        return False      # (the range query that gathers its own centroids serves these chunk shapes)
    main = torch.cuda.current_stream(dev)
    producer = _helper_stream(dev, "producer")
    idx = torch.empty((B, M), dtype=torch.int32, device=dev)
    new_xyz = torch.empty((B, M, 3), dtype=torch.float32, device=dev)
    idx_a = torch.empty((B, M, ga.nsample), dtype=torch.int32, device=dev)
    idx_b = torch.empty((B, M, gb.nsample), dtype=torch.int32, device=dev)
    zeros = _zeroed_ints(dev, 2 * B)
    progress, timed_out = zeros[:B], zeros[B:]
    _TIMEOUT_FLAGS.append(timed_out)
    del _TIMEOUT_FLAGS[:-64]
    start = torch.cuda.Event()
    start.record(main)
    producer.wait_event(start)
    with torch.cuda.stream(producer), torch.no_grad():
        work = _ext.fps_publish(xyz, None, idx, progress, presort=True)
    for t in (xyz, idx, progress) + (() if work is None else (work,)):
        t.record_stream(producer)
    with torch.no_grad():
        for j0, j1 in bounds:
            _ext.wait_progress(progress, j1, timed_out, patient=(j1 == M))
            _ext.ball_query_full2_range(ga.radius, gb.radius, xyz, new_xyz, idx_a, idx_b, j0, j1 - j0,
                                        full_range_if=timed_out if j1 == M else None, gather_idx=idx)
    layer._presampled = (idx, None, xyz, new_xyz)
    layer._preball = (new_xyz, (idx_a, idx_b))
    return True


def prefetch_first_layer(layers_or_layer, xyz):
    """For a TRAINING LOOP that already holds its next batch: start layer 0's sampling (D-FPS) and ball queries for `xyz` now,
    on a side stream, so that they run beside whatever the caller enqueues next -- typically the previous step's backward.
    FPS is a serial chain on one CU per scene (1.7 ms at 8 x 16384 -> 4096) that depends on nothing but the coordinates;
    the next forward over the SAME tensor object (`run_sa_layers`, the backbones) picks the results up instead of sampling
    again.  Exact: same kernels, same picks and rows.  Returns False (and does nothing) when layer 0 does not qualify.

        for batch in loader:                       # reference loop: tools/train_utils/train_utils.py
            out = model(batch); loss = criterion(out)
            sa_stack.prefetch_first_layer(model.SA_modules, next_batch_xyz)    # <- one line
            loss.backward(); optimizer.step()
    """
    layer = layers_or_layer[0] if isinstance(layers_or_layer, (list, tuple, torch.nn.ModuleList)) else layers_or_layer
    if not (xyz.is_cuda and layer.training):
        return False
    dev = xyz.device
    main = torch.cuda.current_stream(dev)
    side = _helper_stream(dev, "prefetch")
    ready = torch.cuda.Event()
    ready.record(main)
    xyz.record_stream(side)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        ok = _streamed_first_layer_queries(layer, xyz)
        if ok:
            done = torch.cuda.Event()
            done.record(side)
            idx, _, src, new_xyz = layer._presampled
            layer._presampled = (idx, done, src, new_xyz)
    return bool(ok)


_ZERO_POOL = {}


def _zeroed_ints(device, count):
    """`count` zeroed int32 that nobody has written yet: slices of a pre-zeroed buffer (one fill launch per 4096 ints handed
    out instead of one per pass, which sat in front of the FPS producer)."""
    if torch.cuda.is_current_stream_capturing():
        # a graph replays its launches on the SAME memory: the zeroes must be produced inside the graph, every replay
        return torch.zeros((count,), dtype=torch.int32, device=device)
    key = (device.type, device.index, _lib.raw_stream(device))
    buf, used = _ZERO_POOL.get(key, (None, 1 << 30))
    if buf is None or used + count > buf.numel():
        buf, used = torch.zeros((max(4096, count),), dtype=torch.int32, device=device), 0
    _ZERO_POOL[key] = (buf, used + count)
    return buf[used:used + count]


def _redo_layer(layer, plan, tail, xyz, new_xyz, features, idx_a, idx_b, out, timed_out):
    """The whole layer once more (behind the repaired centroids), every launch predicated on the device flags `timed_out`
    (see _streamed_first_layer)."""
    from . import fused as _fused
    from . import pointnet2_batch_cuda as _ext
    ga, gb = layer.groupers
    M = new_xyz.shape[1]
    if max(ga.nsample, gb.nsample) > 32:   # units of one centroid meet through an atomic max: the repair needs zeros again
        out.mul_((1 - timed_out[:1]).to(out.dtype))
    _ext.ball_query_full2_range(ga.radius, gb.radius, xyz, new_xyz, idx_a, idx_b, 0, M, run_if=timed_out)
    off = 0
    for ix, packed in zip((idx_a, idx_b), plan):
        _fused.group_mlp_pool(xyz, new_xyz, features, ix, packed, out, off, 0, M, run_if=timed_out)
        off += packed.c3_real
    if tail is not None:
        tail.run(0, M, run_if=timed_out)


def _prefetched_for(layer, xyz):
    """True if prefetch_first_layer (or an earlier call) already sampled and queried `xyz` for this layer."""
    pre, ball = getattr(layer, "_presampled", None), getattr(layer, "_preball", None)
    return pre is not None and ball is not None and len(pre) > 3 and pre[2] is xyz and ball[0] is pre[3]


def _is_plain_dfps(layer, n_in):
    """True if the layer's centroids are ONE D-FPS pick sequence over its whole input."""
    types = getattr(layer, "sample_type_list", None)
    if not types or len(types) != 1 or layer.sample_range_list[0] != -1:
        return False
    t = types[0]
    return ('D-FPS' in t or 'DFS' in t) and not ('cls' in t or 'ctr' in t or 'ss' in t) and n_in > layer.npoint_list[0] > 0


def run_sa_layers(layers, xyz, features, stds=None, overlap=True, stream_first_layer=True):
    """IASSD_backbone.py:128-134 for SA layers: -> list of (new_xyz, new_features, cls, sampled_idx).
    With overlap (inference on a GPU), layer k+1's D-FPS is issued on a side stream the moment layer k's
    new_xyz exists; results are identical, only the schedule changes.
    stream_first_layer: additionally let layer 0's grouping/MLP consume the FPS output while FPS runs
    (_streamed_first_layer; exact, same kernels, only the schedule differs).  Measured on MI355X at the config-2
    shape: 2.91 -> 2.66 ms per pass."""
    outs = []
    cls_pred = None
    use_overlap = overlap and xyz.is_cuda and not torch.is_grad_enabled()
    prefetch = overlap and xyz.is_cuda   # sampling carries no gradient: the next layer's D-FPS may start early in training too
    for k, layer in enumerate(layers):
        nxt = layers[k + 1] if k + 1 < len(layers) else None
        if use_overlap and k == 0 and cls_pred is None and stream_first_layer:
            res = _streamed_first_layer(layer, nxt, xyz, features, stds)
            if res is not None:
                xyz, features, cls_pred, idx, stds = res
                outs.append((xyz, features, cls_pred, idx))
                continue
        if (k == 0 and stream_first_layer and prefetch and not use_overlap and cls_pred is None
                and STREAM_TRAINING_QUERIES and layer.training and not _prefetched_for(layer, xyz)):
            _streamed_first_layer_queries(layer, xyz)
        if prefetch and nxt is not None and _can_prefetch(layer, nxt):
            ordered = _is_plain_dfps(layer, xyz.shape[1])
            layer._on_new_xyz = lambda nx, _n=nxt, _o=ordered: _prefetch_dfps(_n, nx, _o)
        kw = {} if stds is None else {'stds': stds}
        try:
            xyz, features, cls_pred, idx, stds = layer(xyz, features, cls_pred, **kw)
        finally:
            layer._on_new_xyz = None
            if k == 0 and not (getattr(layer, "_presampled", None) is not None and len(layer._presampled) > 3):
                layer._preball = None     # (consumed by the layer; dropped here if it declined or raised -- a prefetch for a
                                          #  FUTURE forward, made inside this one, carries its own `_presampled` and stays)
        outs.append((xyz, features, cls_pred, idx))
    return outs


def pipelined_bench(step, steps, dev, in_flight=2, scenes=8, fenced=True):
    """Time `steps` complete, independent passes issued round-robin on `in_flight` HIP streams (bench.py's informational
    `pipelined` object; `value` stays the strictly sequential figure) -> dict with elapsed_s.
    fenced: every slot's FPS chain runs on compute units of its own and everything else of every slot is kept off them
    (CuFence); unfenced, a second pass in flight is SLOWER per pass than one alone (profiles/README.md, round 2)."""
    import time
    if fenced:
        fences = [CuFence(dev, slot=k, slots=in_flight, scenes=scenes) for k in range(in_flight)]
        streams = [f.stream(f.rest_bits) for f in fences]
        for f, s_ in zip(fences, streams):
            _FENCES[(dev.index, s_.cuda_stream)] = f
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream(dev))
    keep = []
    for i in range(2 * in_flight):            # warm every stream's helper streams / caches
        with torch.cuda.stream(streams[i % in_flight]):
            keep.append(step())
    torch.cuda.synchronize()
    keep.clear()
    t1 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % in_flight]):
            keep.append(step())
            if len(keep) > 2 * in_flight:
                keep.pop(0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
    if fenced:
        for s_ in streams:
            _FENCES.pop((dev.index, s_.cuda_stream), None)
    return {"batches_in_flight": in_flight, "cu_fenced": bool(fenced), "unit": "points/s", "ms_per_step": 1e3 * el / steps,
            "elapsed_s": el, "last_outputs": keep[-1],
            "note": "same complete, independent passes issued round-robin on several HIP streams" +
                    (", each pass's FPS chain on compute units of its own (hipExtStreamCreateWithCUMask)" if fenced else "") +
                    "; informational, `value` above is the strictly sequential figure"}
