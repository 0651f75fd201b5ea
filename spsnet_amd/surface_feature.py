"""Dense edge-convolution surface features over radius neighbourhoods, on the gfx950 ops.

Mirror of the reference's pcdet/ops/pointnet2/pointnet2_batch/surface_feature.py (FCLayer :8-27,
Aggregator :29-43, DenseEdgeConv :45-116, FeatureExtraction :119-187): same class names,
constructor keywords, forward signatures and state_dict keys (`transforms.{i}.linear.*`,
`convs.{i}.layer_first.linear.*`, `convs.{i}.layers.{k}.linear.*`, `convs.{i}.layer_last.linear.*`), used by
PAGNet_backbone.py:30-31,154-157.  The "k-NN" of the reference is a radius query: the first `knn`
points in index order inside radius 0.8 (QueryAndGroup(radius, knn, use_xyz=False), :55), which is what
`sps_query_and_group` computes in one fused launch pair.

Reference quirk kept on purpose: with dynamic_graph=True (the default) the d-channel feature tensor is
passed as `pos`, and the ball query reads its memory as packed (x,y,z) triples (:167-170 with
ball_query_gpu.cu:17-19) -- the build hands the same flat buffer to the same indexing, so results agree.

Inference (no gradient needed, CUDA fp32, the configuration FeatureExtraction instantiates) takes two kernels per
convolution: the radius query and `sps_dense_edge_conv` (csrc/dense_edge_conv.hip), which keeps the edge features and
the three dense activations in registers; the transforms run on `sps_linear_rows`.  Training keeps the
differentiable op-by-op form below.
"""
import torch
import torch.nn as nn

from . import pointnet2_utils

# training (gradients wanted) through the fused kernels; False = the differentiable op-by-op form
FUSED_TRAINING = True

_ACTIVATIONS = {
    None: lambda: nn.Identity(),
    'relu': lambda: nn.ReLU(),
    'elu': lambda: nn.ELU(alpha=1.0),
    'lrelu': lambda: nn.LeakyReLU(0.1),
}


class FCLayer(nn.Module):
    """Linear + optional activation (reference :8-27)."""

    def __init__(self, in_features, out_features, bias=True, activation=None):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self.activation = _ACTIVATIONS[activation]()

    def forward(self, x):
        return self.activation(self.linear(x))


class Aggregator(nn.Module):
    """mean / sum / max over one dimension (reference :29-43)."""

    def __init__(self, oper):
        super().__init__()
        assert oper in ('mean', 'sum', 'max')
        self.oper = oper

    def forward(self, x, dim=2):
        if self.oper == 'max':
            return x.max(dim=dim, keepdim=False)[0]
        return x.mean(dim=dim, keepdim=False) if self.oper == 'mean' else x.sum(dim=dim, keepdim=False)


class DenseEdgeConv(nn.Module):
    """Densely connected edge convolution: (x (B,N,d), pos (B,N,*)) -> (B,N,d + L*c) (reference :45-116)."""

    def __init__(self, in_channels, num_fc_layers, growth_rate, radius=0.8, knn=32, aggr='max', activation='relu',
                 relative_feat_only=False):
        super().__init__()
        assert num_fc_layers > 2
        self.in_channels, self.knn = in_channels, knn
        self.num_fc_layers, self.growth_rate = num_fc_layers, growth_rate
        self.relative_feat_only = relative_feat_only
        self.group = pointnet2_utils.QueryAndGroup(radius, knn, use_xyz=False)
        first_in = in_channels if relative_feat_only else 3 * in_channels
        self.layer_first = FCLayer(first_in, growth_rate, bias=True, activation=activation)
        self.layer_last = FCLayer(in_channels + (num_fc_layers - 1) * growth_rate, growth_rate, bias=True,
                                  activation=None)
        self.layers = nn.ModuleList(
            FCLayer(in_channels + i * growth_rate, growth_rate, bias=True, activation=activation)
            for i in range(1, num_fc_layers - 1))
        self.aggr = Aggregator(aggr)

    @property
    def out_channels(self):
        return self.in_channels + self.num_fc_layers * self.growth_rate

    def get_edge_feature(self, x, pos):
        """x (B,N,d) -> edge features (B,N,K,d) or (B,N,K,3d): [centre, neighbour, neighbour - centre]."""
        grouped = self.group(xyz=pos, new_xyz=pos, features=x.permute(0, 2, 1).contiguous())  # (B,d,N,K)
        neigh = grouped.permute(0, 2, 3, 1).contiguous()
        centre = x.unsqueeze(-2).expand_as(neigh)
        if self.relative_feat_only:
            return neigh - centre
        return torch.cat([centre, neigh, neigh - centre], dim=3)

    def _fused(self, x, pos):
        if not x.is_cuda or x.dtype != torch.float32 or pos.dtype != torch.float32:
            return False
        if pointnet2_utils._needs_grad(x, pos, *self.parameters()):
            return False
        from . import fused
        return fused.dense_edge_conv_supported(self)

    def neighbours(self, pos):
        """(B,N,K) int32 table of the radius query the grouper runs (pos read as packed xyz triples)."""
        pos = pos.contiguous()
        return pointnet2_utils.ball_query(self.group.radius, self.knn, pos, pos)

    def _fused_train(self, x, pos):
        """Gradients wanted: the fused forward + recompute-and-backpropagate kernels (fused.DenseEdgeConvTrain)."""
        if not FUSED_TRAINING or not x.is_cuda or x.dtype != torch.float32 or pos.dtype != torch.float32:
            return False
        if not torch.is_grad_enabled():
            return False
        from . import fused
        return fused.dense_edge_conv_supported(self)

    def forward(self, x, pos, idx=None):
        """`idx` (optional) = a neighbour table already computed for the same `pos` (static graphs share one)."""
        if self._fused(x, pos):
            from . import fused
            return fused.dense_edge_conv(self, x, self.neighbours(pos) if idx is None else idx)
        if self._fused_train(x, pos):
            from . import fused
            with torch.no_grad():
                table = self.neighbours(pos.detach()) if idx is None else idx
            lf, l0, ll = self.layer_first.linear, self.layers[0].linear, self.layer_last.linear
            return fused.DenseEdgeConvTrain.apply(x, table, lf.weight, lf.bias, l0.weight, l0.bias, ll.weight, ll.bias, self)
        y = torch.cat([self.layer_first(self.get_edge_feature(x, pos)),
                       x.unsqueeze(-2).repeat(1, 1, self.knn, 1)], dim=-1)
        for layer in self.layers:
            y = torch.cat([layer(y), y], dim=-1)
        y = torch.cat([self.layer_last(y), y], dim=-1)
        return self.aggr(y, dim=-2)


class FeatureExtraction(nn.Module):
    """Stack of (FCLayer transform, DenseEdgeConv); (B,N,3) -> (B,N,out_channels) (reference :119-187)."""

    def __init__(self, in_channels=3, dynamic_graph=True, conv_channels=24, num_convs=4, conv_num_fc_layers=3,
                 conv_growth_rate=12, conv_knn=16, conv_aggr='max', activation='relu'):
        super().__init__()
        self.in_channels, self.dynamic_graph, self.num_convs = in_channels, dynamic_graph, num_convs
        self.transforms = nn.ModuleList()
        self.convs = nn.ModuleList()
        for i in range(num_convs):
            self.transforms.append(FCLayer(in_channels, conv_channels, bias=True,
                                           activation=None if i == 0 else activation))
            conv = DenseEdgeConv(conv_channels, num_fc_layers=conv_num_fc_layers, growth_rate=conv_growth_rate,
                                 knn=conv_knn, aggr=conv_aggr, activation=activation, relative_feat_only=(i == 0))
            self.convs.append(conv)
            in_channels = conv.out_channels

    @property
    def out_channels(self):
        return self.convs[-1].out_channels

    @staticmethod
    def _transform(fc, x):
        if x.is_cuda and x.dtype == torch.float32 and not pointnet2_utils._needs_grad(x, *fc.parameters()):
            from . import fused
            if fused.linear_rows_supported(fc):
                return fused.linear_rows(fc, x)
        elif FUSED_TRAINING and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled():
            from . import fused
            if fused.linear_rows_supported(fc) and fc.linear.weight.dtype == torch.float32:
                return fused.LinearRowsTrain.apply(x, fc.linear.weight, fc.linear.bias, isinstance(fc.activation, nn.ReLU))
        return fc(x)

    def dynamic_graph_forward(self, x):
        for transform, conv in zip(self.transforms, self.convs):
            x = self._transform(transform, x)
            x = conv(x, x)
        return x

    def static_graph_forward(self, pos):
        x = pos
        idx = None
        for transform, conv in zip(self.transforms, self.convs):
            x = self._transform(transform, x)
            if conv._fused(x, pos) or conv._fused_train(x, pos):
                # every convolution queries the same positions with the same radius and K: one table serves all
                if idx is None:
                    idx = conv.neighbours(pos)
                x = conv(x, pos, idx)
            else:
                x = conv(x, pos)
        return x

    def forward(self, x):
        return self.dynamic_graph_forward(x) if self.dynamic_graph else self.static_graph_forward(x)
