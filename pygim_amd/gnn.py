"""GCN / SAGE / GIN inference stacks whose aggregation runs on the HIP backend.

Same structure and call pattern as the reference's models/models.py:12-131 and the three
``message_and_aggregate`` conv layers (pyg_gcn_conv.py:116-137, pyg_gin_conv.py:74-101,
pyg_sage_conv.py:122-155), written on plain torch.nn (torch_geometric is not installable here):

    Linear -> BN -> ReLU -> [conv -> BN -> ReLU] x L -> Linear          (dropout is identity in eval)
    GCNConv : lin(x) (no bias) -> aggregate -> + bias        (no degree normalisation in the reference)
    SAGEConv: lin_l(aggregate(x)) + lin_r(x)                  (sum aggregation through adj_t.mul)
    GINConv : nn((1 + eps) * x + aggregate(x)),  nn = Linear -> BN -> ReLU -> Linear  (PyG MLP([h, h, h]))
with aggregate = quantise -> adj_t.mul -> dequantise (pygim_amd/quantize.py).  ``adj_t`` is a
SparseTensor (cpu path), a backend_pim SparseTensorCOO, or a pygim_amd.dist.RowShardAdj (multi-GPU:
x and the result are then this rank's row block).
"""
import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Linear, ReLU, Sequential

from .quantize import message_and_aggregate


class GCNConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, **_):
        super().__init__()
        self.lin = Linear(in_channels, out_channels, bias=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, adj_t):
        out = message_and_aggregate(adj_t, self.lin(x))
        return out if self.bias is None else out + self.bias


class SAGEConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, **_):
        super().__init__()
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False)

    def forward(self, x, adj_t):
        return self.lin_l(message_and_aggregate(adj_t, x)) + self.lin_r(x)


class GINConv(torch.nn.Module):
    def __init__(self, nn, eps=0.0):
        super().__init__()
        self.nn = nn
        self.register_buffer("eps", torch.tensor([float(eps)]))

    def forward(self, x, adj_t):
        return self.nn(message_and_aggregate(adj_t, x) + (1 + self.eps) * x)


def folded_epilogue(conv_bias, bn):
    """bias + eval-mode BatchNorm as ONE affine map per feature: bn(y + bias) = a * y + b with
    a = weight / sqrt(running_var + eps), b = (bias - running_mean) * a + bn.bias"""
    a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    bias = conv_bias if conv_bias is not None else torch.zeros_like(bn.running_mean)
    return a, (bias - bn.running_mean) * a + bn.bias


class _Stack(torch.nn.Module):
    fuse_post = False  # True: a GCN layer's "+ bias -> BatchNorm (eval) -> ReLU" runs in the aggregation's last store

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, make_conv):
        super().__init__()
        self.ln1 = Linear(in_channels, hidden_channels)
        self.bn0 = BatchNorm1d(hidden_channels)
        self.convs = torch.nn.ModuleList([make_conv(hidden_channels) for _ in range(num_layers)])
        self.bns = torch.nn.ModuleList([BatchNorm1d(hidden_channels) for _ in range(num_layers)])
        self.ln2 = Linear(hidden_channels, out_channels)
        self.dropout = dropout

    def forward(self, x, adj_t, edge_attr=None):
        x = F.dropout(F.relu(self.bn0(self.ln1(x))), p=self.dropout, training=self.training)
        for conv, bn in zip(self.convs, self.bns):
            if (self.fuse_post and not self.training and isinstance(conv, GCNConv) and x.is_cuda
                    and hasattr(adj_t, "mul_quantized") and not getattr(adj_t, "row_sharded", False)
                    and adj_t.dtype in (torch.int8, torch.int16, torch.int32, torch.float32)):
                # same mathematics as the three torch ops below, one rounding sequence instead of three passes over [N, h]
                a, b = folded_epilogue(conv.bias, bn)
                x, _ = adj_t.mul_quantized(conv.lin(x), post=(a, b, True))
                continue
            x = F.dropout(F.relu(bn(conv(x, adj_t))), p=self.dropout, training=self.training)
        return self.ln2(x)


class GCN(_Stack):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, dropout=0.5):
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, dropout, lambda h: GCNConv(h, h))


class SAGE(_Stack):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, dropout=0.5):
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, dropout, lambda h: SAGEConv(h, h))


class GIN(_Stack):
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, dropout=0.5):
        mlp = lambda h: Sequential(Linear(h, h), BatchNorm1d(h), ReLU(), Linear(h, h))
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, dropout, lambda h: GINConv(mlp(h)))
