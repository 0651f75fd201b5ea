"""IA-SSD / SPSNet point backbones over the gfx950 set-abstraction ops.

Mirror of the reference's pcdet/models/backbones_3d/IASSD_backbone.py (IASSD_Backbone :7-191) and
PAGNet_backbone.py (PAGNet_Backbone :7-212): same class names, constructor `(model_cfg, num_class, input_channels)`,
YAML keys (SA_CONFIG: NPOINT_LIST, SAMPLE_RANGE_LIST, SAMPLE_METHOD_LIST, RADIUS_LIST, NSAMPLE_LIST, MLPS, LAYER_TYPE,
DILATED_GROUP, AGGREGATION_MLPS, CONFIDENCE_MLPS, LAYER_INPUT, CTR_INDEX, MAX_TRANSLATE_RANGE, USE_SURFACE --
tools/cfgs/kitti_models/IA-SSD.yaml:33-56, SPSNet.yaml:39-69), state_dict keys (`SA_modules.{k}...`, `SF_extract...`)
and the keys written to `batch_dict` (:170-178 / :189-197).  `model_cfg` may be the reference's EasyDict or a plain dict.

What differs is the schedule, not the arithmetic (inference on a GPU only; training runs layer by layer):
  * layer 0 consumes its farthest-point picks while FPS is still running and layer 1's FPS starts as soon as layer 0's
    centroids exist (sa_stack._streamed_first_layer / _prefetch_dfps);
  * PAGNet's surface features (FeatureExtraction over all N points, PAGNet_backbone.py:154-157) run on a side stream
    beside layer 0's FPS, which occupies one CU per scene.
The debugging branch that dumps sampled points to a hard-coded path (SAVE_SAMPLE_LIST, :199-212) is not reproduced.
"""
import copy

import torch
import torch.nn as nn

from . import _lib, pointnet2_modules, pointnet2_utils, sa_stack, streams, surface_feature




def equal_counts_check(batch_idx, batch_size):
    """The reference's `assert xyz_batch_cnt.min() == xyz_batch_cnt.max()` (IASSD_backbone.py:109-113,
    stability_generate/model.py:135-139) without stalling the queue: the per-scene counts are computed asynchronously and
    the returned callable raises AssertionError exactly as the reference would.  On a GPU the callable reads the verdict
    on a side stream that waits only for the count kernels, so the caller can enqueue its layers BEFORE blocking -- the
    reference's B `.sum()` calls + assert (and the bincount this replaced: four device-to-host round trips) left the GPU
    idle for ~0.25 ms per forward at 8 x 16 384."""
    def count():
        scenes = torch.arange(batch_size, device=batch_idx.device, dtype=batch_idx.dtype)
        counts = (batch_idx.view(1, -1) == scenes.view(-1, 1)).sum(dim=1)
        return counts.min() == counts.max()

    if not batch_idx.is_cuda:
        ok = count()

        def verdict():
            assert bool(ok), "scenes of unequal size"
            return True
        return verdict
    # the count itself (seven small launches, ~60 us at 8 x 16 384) runs on the side stream too: the caller's stream goes
    # straight to its first layer, whose FPS leaves 248 compute units idle for them
    dev = batch_idx.device
    side = streams.helper(dev, torch.cuda.current_stream(dev), "check")   # (cached per device / pass / role in streams.py)
    entry = torch.cuda.Event()
    entry.record(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        side.wait_event(entry)
        ok = count()
    batch_idx.record_stream(side)

    capturing = torch.cuda.is_current_stream_capturing()
    if capturing:
        torch.cuda.current_stream(dev).wait_stream(side)     # (a capture ends with every forked stream joined back)

    def verdict():
        if capturing:
            # a stream capture cannot read device data on the host: the verdict stays on the device (spsnet_amd.graphs puts it
            # into the replayed batch_dict as 'scene_sizes_equal'); the eager warm-up passes in front of the capture assert
            return ok
        with torch.cuda.stream(side):
            good = bool(ok)          # (blocks the host until the count kernels are done, nothing else)
        assert good, "scenes of unequal size"
        return True
    return verdict


class _Cfg:
    """Attribute + .get() access over a dict or an EasyDict-like object (the reference reads both ways)."""

    def __init__(self, obj):
        self._o = obj

    def _raw(self, key):
        if isinstance(self._o, dict):
            return self._o[key]
        return getattr(self._o, key)

    def __getattr__(self, key):
        if key.startswith('_'):
            raise AttributeError(key)
        try:
            v = self._raw(key)
        except KeyError:
            raise AttributeError(key)
        return _Cfg(v) if isinstance(v, dict) or (hasattr(v, 'keys') and hasattr(v, 'get')) else v

    def get(self, key, default=None):
        try:
            return self.__getattr__(key)
        except AttributeError:
            return default


class _PointBackbone(nn.Module):
    """Common body: IASSD_backbone.py:10-84 / PAGNet_backbone.py:10-92 (constructor), :93-178 / :102-197 (forward)."""

    _surface = False  # PAGNet: FeatureExtraction + stds

    def __init__(self, model_cfg, num_class, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels - 3
        channel_out_list = [channel_in]
        self.num_points_each_layer = []

        sa_config = _Cfg(model_cfg).SA_CONFIG
        self.layer_types = list(sa_config.LAYER_TYPE)
        self.ctr_idx_list = list(sa_config.CTR_INDEX)
        self.layer_inputs = list(sa_config.LAYER_INPUT)
        self.aggregation_mlps = sa_config.get('AGGREGATION_MLPS', None)
        self.confidence_mlps = sa_config.get('CONFIDENCE_MLPS', None)
        self.max_translate_range = sa_config.get('MAX_TRANSLATE_RANGE', None)
        if self._surface and sa_config.get('USE_SURFACE', False):
            self.SF_extract = surface_feature.FeatureExtraction()

        for k in range(len(sa_config.NSAMPLE_LIST)):
            src = self.layer_inputs[k]
            channel_in = channel_out_list[src[-1] if isinstance(src, list) else src]
            if self.layer_types[k] == 'SA_Layer':
                mlps = [[channel_in] + list(m) for m in sa_config.MLPS[k]]
                channel_out = sum(m[-1] for m in mlps)
                aggregation_mlp = None
                if self.aggregation_mlps and self.aggregation_mlps[k]:
                    aggregation_mlp = list(self.aggregation_mlps[k])
                    channel_out = aggregation_mlp[-1]
                confidence_mlp = None
                if self.confidence_mlps and self.confidence_mlps[k]:
                    confidence_mlp = list(self.confidence_mlps[k])
                self.SA_modules.append(pointnet2_modules.PointnetSAModuleMSG_WithSampling(
                    npoint_list=sa_config.NPOINT_LIST[k], sample_range_list=sa_config.SAMPLE_RANGE_LIST[k],
                    sample_type_list=sa_config.SAMPLE_METHOD_LIST[k], radii=sa_config.RADIUS_LIST[k],
                    nsamples=sa_config.NSAMPLE_LIST[k], mlps=mlps, use_xyz=True,
                    dilated_group=sa_config.DILATED_GROUP[k], aggregation_mlp=aggregation_mlp,
                    confidence_mlp=confidence_mlp, num_class=self.num_class, **self._sampler_kwargs(sa_config, k)))
            elif self.layer_types[k] == 'Vote_Layer':
                self.SA_modules.append(pointnet2_modules.Vote_layer(
                    mlp_list=sa_config.MLPS[k], pre_channel=channel_out_list[self.layer_inputs[k]],
                    max_translate_range=self.max_translate_range))
                # channel_out keeps the previous layer's value, as in the reference (:76-84)
            if hasattr(self, 'SF_extract') and k == 3:
                channel_out += 60  # the voting layer sees [surface features | layer-3 features] (PAGNet_backbone.py:88-89)
            channel_out_list.append(channel_out)
        self.num_point_features = channel_out

    def _sampler_kwargs(self, sa_config, k):
        return {}

    def break_up_pc(self, pc):
        batch_idx = pc[:, 0]
        xyz = pc[:, 1:4].contiguous()
        features = (pc[:, 4:].contiguous() if pc.size(-1) > 4 else None)
        return batch_idx, xyz, features

    # ------------------------------------------------------------------------------------------------------------------
    def _sa_layer(self, i, xyz_input, feature_input, cls_pred, ctr_xyz, stds, fast, beside_fps=None):
        """One SA layer; on the fast path layer 0 is streamed against its own FPS and the next plain D-FPS layer is
        started early.  -> (li_xyz, li_features, li_cls_pred, sampled_idx, stds)"""
        layer = self.SA_modules[i]
        nxt = None
        if (xyz_input.is_cuda and i + 1 < len(self.SA_modules) and self.layer_types[i + 1] == 'SA_Layer'
                and self.layer_inputs[i + 1] == i + 1 and self.ctr_idx_list[i + 1] == -1):  # sampling has no gradient
            nxt = self.SA_modules[i + 1]
        if fast and i == 0 and cls_pred is None and ctr_xyz is None and self.layer_inputs[0] == 0:
            res = sa_stack._streamed_first_layer(layer, nxt, xyz_input, feature_input, stds, after_producer=beside_fps)
            if res is not None:
                return res
        if (i == 0 and layer.training and xyz_input.is_cuda and cls_pred is None and ctr_xyz is None
                and self.layer_inputs[0] == 0 and sa_stack.STREAM_TRAINING_QUERIES
                and not sa_stack._prefetched_for(layer, xyz_input)):
            sa_stack._streamed_first_layer_queries(layer, xyz_input)   # training: the ball queries run beside the FPS
        if beside_fps is not None:
            beside_fps()   # no streamed layer 0: at least ahead of its kernels
        if nxt is not None and ctr_xyz is None and sa_stack._can_prefetch(layer, nxt):
            ordered = sa_stack._is_plain_dfps(layer, xyz_input.shape[1])
            layer._on_new_xyz = lambda nx, _n=nxt, _o=ordered: sa_stack._prefetch_dfps(_n, nx, _o)
        kw = {}
        if self._surface:
            kw['stds'] = stds
        try:
            return layer(xyz_input, feature_input, cls_pred, ctr_xyz=ctr_xyz, **kw)
        finally:
            layer._on_new_xyz = None
            if not (getattr(layer, "_presampled", None) is not None and len(layer._presampled) > 3):
                layer._preball = None

    def prefetch_sampling(self, batch_dict):
        """For a training loop that already holds its NEXT batch: start layer 0's FPS and ball queries for it now, on a side
        stream (sa_stack.prefetch_first_layer), e.g. right before `loss.backward()` of the current step; the forward over
        the same `batch_dict['points']` tensor picks the results up.  Exact; returns False when layer 0 does not qualify."""
        points = batch_dict['points']
        if not (points.is_cuda and self.training and self.layer_inputs[0] == 0 and self.ctr_idx_list[0] == -1):
            return False
        xyz = points[:, 1:4].contiguous().view(batch_dict['batch_size'], -1, 3)
        if not sa_stack.prefetch_first_layer(self.SA_modules[0], xyz):
            return False
        self._prefetched = (points, xyz)
        return True

    def forward(self, batch_dict):
        """batch_dict: batch_size, points (B*N, 4 + C) [batch_idx, x, y, z, ...] (+ stds for PAGNet) -> batch_dict with
        ctr_offsets, centers, centers_origin, centers_features, ctr_batch_idx, encoder_xyz, encoder_coords,
        sa_ins_preds, encoder_features."""
        batch_size = batch_dict['batch_size']
        points = batch_dict['points']
        batch_idx, xyz, features = self.break_up_pc(points)
        pre, self._prefetched = getattr(self, "_prefetched", None), None
        if pre is not None and pre[0] is points:
            xyz = pre[1]          # prefetch_sampling already sampled / queried THIS tensor object for layer 0
        stds = batch_dict.get('stds', None) if self._surface else None

        # every scene must hold the same number of points (reference :109-113); on the GPU the verdict is read once the
        # whole forward is in the queue, on a stream of its own (equal_counts_check)
        counts_ok = equal_counts_check(batch_idx, batch_size)
        if xyz.dim() != 3:        # (a prefetched tensor is (B, N, 3) already, and must stay the SAME object)
            xyz = xyz.view(batch_size, -1, 3)
        features = (features.view(batch_size, -1, features.shape[-1]).permute(0, 2, 1).contiguous()
                    if features is not None else None)

        fast = xyz.is_cuda and not torch.is_grad_enabled()
        encoder_xyz, encoder_features, sa_ins_preds = [xyz], [features], []
        encoder_coords = [torch.cat([batch_idx.view(batch_size, -1, 1), xyz], dim=-1)]
        sample_list = []
        surface = None
        surface_done = None
        pending_gathers = []   # fast path: the per-layer gathers of the surface features are deferred to their consumer
        xyz_ready = None
        if fast and self._surface and hasattr(self, 'SF_extract'):
            xyz_ready = torch.cuda.Event()
            xyz_ready.record(torch.cuda.current_stream(xyz.device))

        def launch_surface():
            # all-N surface features on a side stream, enqueued right after layer 0's FPS kernel (which occupies one CU
            # per scene for ~1.9 ms) and before the chunk pipeline that consumes its picks
            nonlocal surface, surface_done
            side = _surface_stream(xyz.device)
            xyz.record_stream(side)
            with torch.cuda.stream(side):
                side.wait_event(xyz_ready)
                surface = self.SF_extract(xyz).permute(0, 2, 1).contiguous()
                surface_done = torch.cuda.Event()
                surface_done.record(side)

        def gathered_surface():
            nonlocal surface, surface_done
            if surface_done is not None:
                torch.cuda.current_stream(xyz.device).wait_event(surface_done)
                surface.record_stream(torch.cuda.current_stream(xyz.device))
                surface_done = None
            for sampled in pending_gathers:
                surface = pointnet2_utils.gather_operation(surface, sampled)
            pending_gathers.clear()
            return surface

        li_cls_pred = None
        centers = centers_origin = ctr_offsets = None
        for i in range(len(self.SA_modules)):
            xyz_input = encoder_xyz[self.layer_inputs[i]]
            feature_input = encoder_features[self.layer_inputs[i]]
            if self.layer_types[i] == 'SA_Layer':
                ctr_xyz = encoder_xyz[self.ctr_idx_list[i]] if self.ctr_idx_list[i] != -1 else None
                hook = launch_surface if (fast and i == 0 and xyz_ready is not None) else None
                li_xyz, li_features, li_cls_pred, sampled_idx_list, stds = self._sa_layer(
                    i, xyz_input, feature_input, li_cls_pred, ctr_xyz, stds, fast, hook)
                sample_list.append(sampled_idx_list)
                if self._surface and hasattr(self, 'SF_extract') and i <= 4:
                    if fast:
                        pending_gathers.append(sampled_idx_list)
                    else:
                        if i == 0:
                            surface = self.SF_extract(xyz).permute(0, 2, 1).contiguous()
                        surface = pointnet2_utils.gather_operation(surface, sampled_idx_list)
            elif self.layer_types[i] == 'Vote_Layer':
                kw = {'center_surface_futures': gathered_surface() if fast else surface} if self._surface else {}
                li_xyz, li_features, xyz_select, ctr_offsets = self.SA_modules[i](xyz_input, feature_input, **kw)
                centers = li_xyz
                centers_origin = xyz_select
                origin_idx = batch_idx.view(batch_size, -1)[:, :centers_origin.shape[1]]
                encoder_coords.append(torch.cat([origin_idx[..., None].float(), centers_origin.view(batch_size, -1, 3)], dim=-1))
            encoder_xyz.append(li_xyz)
            li_batch_idx = batch_idx.view(batch_size, -1)[:, :li_xyz.shape[1]]
            encoder_coords.append(torch.cat([li_batch_idx[..., None].float(), li_xyz.view(batch_size, -1, 3)], dim=-1))
            encoder_features.append(li_features)
            if li_cls_pred is not None:
                cls_idx = batch_idx.view(batch_size, -1)[:, :li_cls_pred.shape[1]]
                sa_ins_preds.append(torch.cat([cls_idx[..., None].float(),
                                               li_cls_pred.view(batch_size, -1, li_cls_pred.shape[-1])], dim=-1))
            else:
                sa_ins_preds.append([])

        sizes_equal = counts_ok()   # everything is in the queue: blocking here starves nothing
        if isinstance(sizes_equal, torch.Tensor):      # (under stream capture the verdict stays on the device: spsnet_amd.graphs)
            batch_dict['scene_sizes_equal'] = sizes_equal
        ctr_batch_idx = batch_idx.view(batch_size, -1)[:, :li_xyz.shape[1]].contiguous().view(-1)
        col = ctr_batch_idx[:, None].float()
        batch_dict['ctr_offsets'] = torch.cat((col, ctr_offsets.contiguous().view(-1, 3)), dim=1)
        batch_dict['centers'] = torch.cat((col, centers.contiguous().view(-1, 3)), dim=1)
        batch_dict['centers_origin'] = torch.cat((col, centers_origin.contiguous().view(-1, 3)), dim=1)
        last = encoder_features[-1]
        batch_dict['centers_features'] = last.permute(0, 2, 1).contiguous().view(-1, last.shape[1])
        batch_dict['ctr_batch_idx'] = ctr_batch_idx
        batch_dict['encoder_xyz'] = encoder_xyz
        batch_dict['encoder_coords'] = encoder_coords
        batch_dict['sa_ins_preds'] = sa_ins_preds
        batch_dict['encoder_features'] = encoder_features
        return batch_dict


def _surface_stream(device):
    return streams.helper(device, torch.cuda.current_stream(device), "surface")


class IASSD_Backbone(_PointBackbone):
    """Backbone for IA-SSD (reference IASSD_backbone.py:7-191)."""


class PAGNet_Backbone(_PointBackbone):
    """Backbone for SPSNet: IA-SSD's stack + stability-driven sampling (`stds`) + surface features for the voting layer
    (reference PAGNet_backbone.py:7-212)."""
    _surface = True

    def _sampler_kwargs(self, sa_config, k):  # stable-sampling balls (PAGNet_backbone.py:76-78)
        ss_r, ss_n = sa_config.get('SS_RADIUS_LIST', None), sa_config.get('SS_NSAMPLE_LIST', None)
        return dict(ss_radii=ss_r[k] if ss_r is not None else None, ss_nsamples=ss_n[k] if ss_n is not None else None)


# the shipped KITTI configurations, as plain dicts (tools/cfgs/kitti_models/IA-SSD.yaml:33-56, SPSNet.yaml:39-69)
IASSD_KITTI_CFG = dict(SA_CONFIG=dict(
    NPOINT_LIST=[[4096], [1024], [512], [256], [-1], [256]],
    SAMPLE_RANGE_LIST=[[-1], [-1], [-1], [-1], [-1], [-1]],
    SAMPLE_METHOD_LIST=[['D-FPS'], ['D-FPS'], ['ctr_aware'], ['ctr_aware'], [], []],
    RADIUS_LIST=[[0.2, 0.8], [0.8, 1.6], [1.6, 4.8], [], [], [4.8, 6.4]],
    NSAMPLE_LIST=[[16, 32], [16, 32], [16, 32], [], [], [16, 32]],
    MLPS=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 128, 256], [128, 256, 256]], [], [128],
          [[256, 256, 512], [256, 512, 1024]]],
    LAYER_TYPE=['SA_Layer', 'SA_Layer', 'SA_Layer', 'SA_Layer', 'Vote_Layer', 'SA_Layer'],
    DILATED_GROUP=[False, False, False, False, False, False],
    AGGREGATION_MLPS=[[64], [128], [256], [256], [], [512]],
    CONFIDENCE_MLPS=[[], [128], [256], [], [], []],
    LAYER_INPUT=[0, 1, 2, 3, 4, 3],
    CTR_INDEX=[-1, -1, -1, -1, -1, 5],
    MAX_TRANSLATE_RANGE=[3.0, 3.0, 2.0],
))


SPSNET_KITTI_CFG = copy.deepcopy(IASSD_KITTI_CFG)
SPSNET_KITTI_CFG['SA_CONFIG'].update(
    SAMPLE_METHOD_LIST=[['D-FPS'], ['D-FPS'], ['sss_aware'], ['sss_aware'], [], []],
    SS_RADIUS_LIST=[[0.05], [0.2], [], [], [], []],
    SS_NSAMPLE_LIST=[[16], [16], [], [], [], [1]],
    USE_SURFACE=True,
)
SPSNET_KITTI_CFG['SA_CONFIG']['MLPS'][1] = [[124, 64, 128], [124, 96, 128]]


SEARCH_BESIDE_ENCODER = True  # PointNet2MSG inference: the decoder's 3-NN searches run beside the encoder
STREAM_FIRST_LAYER = True     # PointNet2MSG inference: layer 0 consumes its FPS picks while FPS runs (tests switch it off to compare)

# tools/cfgs/kitti_models/pointrcnn.yaml: BACKBONE_3D of PointRCNN (the reference's other point-based detector)
POINTRCNN_KITTI_CFG = dict(
    NAME='PointNet2MSG',
    SA_CONFIG=dict(
        NPOINTS=[4096, 1024, 256, 64],
        RADIUS=[[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
        NSAMPLE=[[16, 32], [16, 32], [16, 32], [16, 32]],
        MLPS=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]],
              [[256, 256, 512], [256, 384, 512]]]),
    FP_MLPS=[[128, 128], [256, 256], [512, 512], [512, 512]])


class PointNet2MSG(nn.Module):
    """The encoder / decoder backbone of PointRCNN (pcdet/models/backbones_3d/pointnet2_backbone.py:9-100): a chain of
    multi-scale-grouping SA layers with plain D-FPS, then feature-propagation layers that carry the coarse features back to
    every input point (3-NN inverse-distance interpolation + a shared MLP over [interpolated | skip] channels).  Same
    constructor arguments, `state_dict` keys (`SA_modules.k.*`, `FP_modules.k.mlp.*`), `num_point_features` and output
    entries (`point_features` (B*N, C), `point_coords` (B*N, 4)) as the reference class; in inference every SA layer runs on
    the fused kernels (one scan for both radii, gather + grouped MLP + max-pool on the matrix cores) and the next layer's
    D-FPS -- a pick sequence over the previous layer's picks -- is the verified identity prefix, started the moment its
    centroids exist."""

    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        cfg = _Cfg(model_cfg)
        sa = cfg.SA_CONFIG
        use_xyz = sa.get('USE_XYZ', True)
        self.SA_modules = nn.ModuleList()
        width = input_channels - 3
        skip_widths = [width]
        for npoint, radii, nsamples, specs in zip(sa.NPOINTS, sa.RADIUS, sa.NSAMPLE, sa.MLPS):
            mlps = [[width] + list(spec) for spec in specs]
            self.SA_modules.append(pointnet2_modules.PointnetSAModuleMSG(
                npoint=npoint, radii=list(radii), nsamples=list(nsamples), mlps=mlps, use_xyz=use_xyz))
            width = sum(spec[-1] for spec in specs)
            skip_widths.append(width)
        fp = [list(m) for m in cfg.FP_MLPS]
        self.FP_modules = nn.ModuleList()
        for k, spec in enumerate(fp):
            coarse = fp[k + 1][-1] if k + 1 < len(fp) else width     # what arrives from the coarser level
            self.FP_modules.append(pointnet2_modules.PointnetFPModule(mlp=[coarse + skip_widths[k]] + spec))
        self.num_point_features = fp[0][-1]

    def _centroids(self, layer, cloud):
        """The layer's D-FPS picks gathered from `cloud` (pointnet2_modules.py:59-63) -- from the early start on a side stream
        when the previous layer's centroids were `cloud` (see forward), else sampled here."""
        pre, layer._presampled = getattr(layer, "_presampled", None), None
        idx = None
        if pre is not None and pre[2] is cloud and pre[0].shape[1] == layer.npoint:
            idx, done = pre[0], pre[1]
            if done is not None:            # (None: produced in line on this stream already)
                main = torch.cuda.current_stream(cloud.device)
                main.wait_event(done)
                idx.record_stream(main)
        if idx is None:
            idx = pointnet2_utils.farthest_point_sample(cloud, layer.npoint)
        if cloud.is_cuda and not (torch.is_grad_enabled() and cloud.requires_grad):
            return pointnet2_modules._ext.gather_xyz(cloud, idx)
        return pointnet2_utils.gather_operation(cloud.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()

    def _search_beside(self, unknown, known):
        """three_nn(unknown, known) on a side stream -> (dist, idx, event)"""
        dev = unknown.device
        main = torch.cuda.current_stream(dev)
        side = sa_stack._helper_stream(dev, "search")
        ready = torch.cuda.Event()
        ready.record(main)
        for t in (unknown, known):
            t.record_stream(side)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            done = torch.cuda.Event()
            done.record(side)
        return dist, idx, done

    def forward(self, batch_dict):
        batch_size = batch_dict['batch_size']
        points = batch_dict['points']
        batch_idx = points[:, 0]
        counts_ok = equal_counts_check(batch_idx, batch_size)
        xyz = points[:, 1:4].contiguous().view(batch_size, -1, 3)
        feats = None
        if points.size(-1) > 4:
            feats = points[:, 4:].contiguous().view(batch_size, -1, points.size(-1) - 4).permute(0, 2, 1).contiguous()
        level_xyz, level_feats = [xyz], [feats]
        early = xyz.is_cuda          # sampling carries no gradient: the early start applies in training too
        # the decoder's 3-NN searches need coordinates only: each starts on a side stream the moment its two levels exist and
        # runs beside the encoder (inference; 0.33 ms of the forward at 8 x 16 384)
        beside = xyz.is_cuda and not torch.is_grad_enabled() and SEARCH_BESIDE_ENCODER
        searches = []
        for k, layer in enumerate(self.SA_modules):
            nxt = self.SA_modules[k + 1] if k + 1 < len(self.SA_modules) else None
            if (k == 0 and early and STREAM_FIRST_LAYER and not torch.is_grad_enabled() and layer.npoint is not None
                    and getattr(layer, "_presampled", None) is None):
                # inference: the first layer's ball queries and grouped MLPs consume its FPS picks while FPS still samples,
                # and the next layer's D-FPS is the verified identity prefix (sa_stack._streamed_first_layer; exact)
                res = sa_stack._streamed_first_layer(_DfpsShim(layer), _DfpsShim(nxt) if nxt is not None and nxt.npoint else None,
                                                     level_xyz[0], level_feats[0])
                if res is not None:
                    level_xyz.append(res[0])
                    level_feats.append(res[1])
                    searches.append(self._search_beside(level_xyz[k], level_xyz[k + 1]) if beside else None)
                    continue
            new_xyz = self._centroids(layer, level_xyz[k]) if layer.npoint is not None else None
            if (early and new_xyz is not None and nxt is not None and nxt.npoint is not None
                    and nxt.npoint < new_xyz.shape[1] and level_xyz[k].shape[1] > layer.npoint):
                # the next layer samples THESE centroids, a D-FPS pick sequence in pick order: its own D-FPS is the identity
                # prefix up to exact distance ties (fps_verify.hip checks it in parallel), started now on a side stream
                # beside this layer's ball queries and grouped MLPs
                sa_stack._prefetch_dfps(_DfpsShim(nxt), new_xyz, fps_ordered=nxt.npoint <= pointnet2_modules._ext.ORDERED_PREFIX_MAX)
            if beside and new_xyz is not None and k < len(self.FP_modules):
                searches.append(self._search_beside(level_xyz[k], new_xyz))    # before the layer's own kernels are enqueued
            else:
                searches.append(None)
            li_xyz, li_feats = layer(level_xyz[k], level_feats[k], new_xyz=new_xyz)
            level_xyz.append(li_xyz)
            level_feats.append(li_feats)
        for k in range(len(self.FP_modules) - 1, -1, -1):
            nn_k = None
            if k < len(searches) and searches[k] is not None:
                dist, idx, done = searches[k]
                main = torch.cuda.current_stream(xyz.device)
                main.wait_event(done)
                dist.record_stream(main)
                idx.record_stream(main)
                nn_k = (dist, idx)
            # (the last module's output leaves as per-point rows: in inference its kernel writes them that way)
            level_feats[k] = self.FP_modules[k](level_xyz[k], level_xyz[k + 1], level_feats[k], level_feats[k + 1], neighbours=nn_k,
                                                point_major_ok=(k == 0))
        if getattr(level_feats[0], "_sps_point_major", False):
            point_features = level_feats[0]
        else:
            point_features = level_feats[0].permute(0, 2, 1).contiguous()
        batch_dict['point_features'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_coords'] = torch.cat((batch_idx[:, None].float(), xyz.view(-1, 3)), dim=1)
        counts_ok()
        return batch_dict


class _DfpsShim:
    """A PointnetSAModuleMSG seen through the interface sa_stack's schedules are written against (the sampling SA module:
    npoint_list / sample_type_list / sample_range_list, groupers, an optional aggregation tail): one plain D-FPS sampler over
    the whole cloud, no tail.  `_presampled` (where the schedules leave an early-started sampling result) lands on the
    wrapped module."""

    aggregation_layer = None
    confidence_layers = None

    def __init__(self, layer):
        object.__setattr__(self, "_layer", layer)
        object.__setattr__(self, "npoint_list", [layer.npoint])
        object.__setattr__(self, "sample_type_list", ['D-FPS'])
        object.__setattr__(self, "sample_range_list", [-1])

    def __getattr__(self, key):                     # groupers, mlps, training, _fused_plan, ...
        return getattr(object.__getattribute__(self, "_layer"), key)

    def __setattr__(self, key, value):
        setattr(self._layer, key, value)

    def _tail(self, pooled, half_out=False):
        return pooled, None


def scaled_cfg(base, npoints):
    """The same stack on a smaller cloud: NPOINT_LIST replaced (entries of -1 kept)."""
    cfg = copy.deepcopy(base)
    cfg['SA_CONFIG']['NPOINT_LIST'] = [[p] for p in npoints]
    return cfg
