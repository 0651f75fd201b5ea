"""The stability generator's inference path: the producer of the per-point `stds` score that drives SPSNet's
stability sampler (pointnet2_modules.py:305-327 consume it; PAGNet_encoding.py:16-26 calls the generator).

Mirror of the reference's stability_generate/model.py: `Surface_PW_feature` (:34-169, one `PointnetSampling` layer with
npoint = N: every point is a centroid), `Encoder_surface_feature` (:172-184), `Object_feat_encoder` (:187-220, kept for
`state_dict` compatibility: it only feeds the training loss) and `Generate_center.forward` in eval mode (:545-588):

    soc_feature = SA layer (ball query r = 0.2 / 0.8 over all N points, grouped MLPs, aggregation) -> (B, N, 64)
    logvar      = fc2(soc_feature)                                                         -> (B, N, latent)
    stds        = sum_k exp(0.5 * logvar_k)                                               -> (B, N)

Same class names, constructor arguments, `model_cfg` keys (stability_generate/cfgs/sf_unc.yaml:52-77) and parameter
names.  The training branch (target assignment through roiaware_pool3d, KL / centre losses, :249-508) is outside the
set-abstraction path and is not reproduced: `forward` raises in training mode.

On the GPU the SA layer is three launches per grouping scale pair -- the two-radius grid ball query
(`sps_ball_query_grid2`: M = N = 16 384 is exactly the case the scan is worst at), one fused MFMA kernel per scale and
the aggregation kernel -- and the head (Linear + exp + sum) is one small fused launch of torch ops.
"""
import torch
import torch.nn as nn

from . import pointnet2_modules
from .backbones import _Cfg, equal_counts_check


class Surface_PW_feature(nn.Module):
    """Point-wise features of every input point (reference :34-169)."""

    def __init__(self, model_cfg, input_channels=4, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels - 3
        channel_out_list = [channel_in]
        self.num_points_each_layer = []
        sa_config = _Cfg(model_cfg)
        self.layer_types = list(sa_config.LAYER_TYPE)
        self.ctr_idx_list = list(sa_config.CTR_INDEX)
        self.layer_inputs = list(sa_config.LAYER_INPUT)
        self.aggregation_mlps = sa_config.get('AGGREGATION_MLPS', None)
        self.confidence_mlps = sa_config.get('CONFIDENCE_MLPS', None)
        self.max_translate_range = sa_config.get('MAX_TRANSLATE_RANGE', None)
        for k in range(len(sa_config.NSAMPLE_LIST)):
            src = self.layer_inputs[k]
            channel_in = channel_out_list[src[-1] if isinstance(src, list) else src]
            mlps = [[channel_in] + list(m) for m in sa_config.MLPS[k]]
            channel_out = sum(m[-1] for m in mlps)
            aggregation_mlp = None
            if self.aggregation_mlps and self.aggregation_mlps[k]:
                aggregation_mlp = list(self.aggregation_mlps[k])
                channel_out = aggregation_mlp[-1]
            self.SA_modules.append(pointnet2_modules.PointnetSampling(
                npoint_list=sa_config.NPOINT_LIST[k], sample_range_list=sa_config.SAMPLE_RANGE_LIST[k],
                sample_type_list=sa_config.SAMPLE_METHOD_LIST[k], radii=sa_config.RADIUS_LIST[k],
                nsamples=sa_config.NSAMPLE_LIST[k], mlps=mlps, use_xyz=True,
                dilated_group=sa_config.DILATED_GROUP[k], aggregation_mlp=aggregation_mlp))
            channel_out_list.append(channel_out)
        self.num_point_features = channel_out

    @staticmethod
    def _columns(points):
        """(sum of points, 1 + 3 + C) rows [scene, x, y, z, features ...] -> (scene, xyz, features or None)."""
        extra = points.shape[-1] - 4
        return points[:, 0], points[:, 1:4].contiguous(), (points[:, 4:].contiguous() if extra > 0 else None)

    def break_up_pc(self, pc):   # (the reference's name for it, :106-110)
        return self._columns(pc)

    def forward(self, batch_dict):
        """Reads `batch_size`, `points`; adds `encoder_xyz`, `encoder_coords`, `sa_ins_preds` and `soc_feature` (B, N, C): the
        keys the reference's forward leaves in the dict (:112-168)."""
        B = batch_dict['batch_size']
        owner, xyz, feats = self._columns(batch_dict['points'])
        counts_ok = equal_counts_check(owner, B)            # verdict read once the layers are in the queue
        owner, xyz = owner.view(B, -1), xyz.view(B, -1, 3)
        if feats is not None:
            feats = feats.view(B, xyz.shape[1], -1).transpose(1, 2).contiguous()
        levels = [(xyz, feats)]                             # the input cloud, then every SA layer's (centroids, features)
        coords = [torch.cat([owner.unsqueeze(-1), xyz], dim=-1)]
        for layer, src, ctr in zip(self.SA_modules, self.layer_inputs, self.ctr_idx_list):
            in_xyz, in_feats = levels[src]
            out_xyz, out_feats, _ = layer(in_xyz, in_feats, None, ctr_xyz=levels[ctr][0] if ctr != -1 else None)
            levels.append((out_xyz, out_feats))
            m = out_xyz.shape[1]
            coords.append(torch.cat([owner[:, :m].unsqueeze(-1).float(), out_xyz.reshape(B, m, 3)], dim=-1))
        counts_ok()
        batch_dict.update(encoder_xyz=[lv[0] for lv in levels], encoder_coords=coords,
                          sa_ins_preds=[[] for _ in self.SA_modules],
                          soc_feature=levels[-1][1].transpose(1, 2).contiguous())
        return batch_dict


class Encoder_surface_feature(nn.Module):
    """Feature -> (mu, logvar) of the latent distribution (reference :172-184)."""

    def __init__(self, input_channels, latent_size=3):
        super().__init__()
        self.fc1 = nn.Linear(input_channels, latent_size)
        self.fc2 = nn.Linear(input_channels, latent_size)

    def forward(self, features):
        mu = self.fc1(features)
        logvar = self.fc2(features)
        dist = torch.distributions.Independent(torch.distributions.Normal(loc=mu, scale=torch.exp(logvar) + 3e-22), 1)
        return dist, mu, logvar


class Object_feat_encoder(nn.Module):
    """Centre regressor of the training loss (reference :187-220); present so that checkpoints load."""

    def __init__(self, model_cfg):
        super().__init__()
        cfg = _Cfg(model_cfg)
        width = int(256 * 0.25)
        self.fc1 = nn.Linear(cfg.PW_FEATURE_DIM + cfg.LATENT_DIM, width)
        self.fc2 = nn.Linear(width, width)
        self.fc_ce1 = nn.Linear(width, width)
        self.fc_ce2 = nn.Linear(width, 3, bias=False)

    def forward(self, x, z):
        x = torch.relu(self.fc1(torch.cat([x, z], dim=-1)))
        feat = torch.relu(self.fc2(x))
        return self.fc_ce2(torch.relu(self.fc_ce1(feat)))


class Generate_center(nn.Module):
    """The stability generator; eval-mode forward writes batch_dict['stds'] (B, N) (reference :222-238, 545-588)."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        cfg = _Cfg(model_cfg)
        self.feature_extract = Surface_PW_feature(cfg.SA_CONFIG._o)
        self.feature_encoder = Encoder_surface_feature(input_channels=cfg.SF_FEATURE_DIM, latent_size=cfg.LATENT_DIM)
        generator = cfg.get('GENERATOR', None)
        if generator is not None:
            self.obj_encoder = Object_feat_encoder(generator._o)
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        if kwargs.get('training', None) is not None:
            self.training = kwargs['training']

    def forward(self, batch_dict, **kwargs):
        if kwargs.get('training', None) is not None:
            self.training = kwargs['training']
        if self.training:
            raise NotImplementedError("Generate_center: only the inference path (stds) is built; the training losses "
                                      "(stability_generate/model.py:249-508) need pcdet's box/roiaware utilities")
        batch_dict = self.feature_extract(batch_dict)
        soc_feature = batch_dict['soc_feature']
        logvarx = self.feature_encoder.fc2(soc_feature)  # the eval branch uses logvar only (:573-575)
        batch_dict['stds'] = torch.sum(logvarx.mul(0.5).exp_(), dim=-1)
        return batch_dict


# the shipped configuration (stability_generate/cfgs/sf_unc.yaml:52-77); GENERATOR as Object_feat_encoder reads it
SF_UNC_CFG = dict(
    SF_FEATURE_DIM=64, LATENT_DIM=8,
    GENERATOR=dict(LATENT_DIM=8, PW_FEATURE_DIM=64),
    SA_CONFIG=dict(
        NPOINT_LIST=[[16384]], SAMPLE_RANGE_LIST=[[-1]], SAMPLE_METHOD_LIST=[['D-FPS']],
        RADIUS_LIST=[[0.2, 0.8]], NSAMPLE_LIST=[[16, 32]], MLPS=[[[16, 16, 32], [32, 32, 64]]],
        LAYER_TYPE=['SA_Layer'], DILATED_GROUP=[False], AGGREGATION_MLPS=[[64]], CONFIDENCE_MLPS=[[]],
        LAYER_INPUT=[0], CTR_INDEX=[-1], MAX_TRANSLATE_RANGE=[3.0, 3.0, 2.0]))
