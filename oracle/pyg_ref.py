"""CPU restatement of the PyG 2.5.2 operators on DeformContact's hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): the checker for the HIP path,
never the product.  **Parity unpinned at the PyG boundary**: ``torch_geometric``
2.5.2 (``/root/reference/environment.yml:76``) is absent from this image, so the
functions below restate its published algorithm from the upstream files named in
each docstring, issuing the same ATen op sequence PyG issues.  Call sites in the
reference that fix which defaults apply: ``models/model.py:39`` (class choice),
``:45,49`` (``conv_layer(in, out)`` - every other ctor arg at its default),
``:71,77`` (``conv(x, edge_index)`` - no edge_weight / edge_attr).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


# --------------------------------------------------------------------------- #
# utils/_scatter.py, utils/loop.py, utils/_softmax.py
# --------------------------------------------------------------------------- #
def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``torch_geometric.utils.scatter(src, index, 0, dim_size, 'sum')``:
    ``src.new_zeros(size).scatter_add_(0, broadcast(index), src)``."""
    size = list(src.shape)
    size[0] = dim_size
    idx = index.view([-1] + [1] * (src.dim() - 1)).expand_as(src)
    return src.new_zeros(size).scatter_add_(0, idx, src)


def scatter_max(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``scatter(..., reduce='max')`` = ``scatter_reduce_('amax', include_self=False)``
    on a zero-initialised output (rows with no entry stay 0)."""
    size = list(src.shape)
    size[0] = dim_size
    idx = index.view([-1] + [1] * (src.dim() - 1)).expand_as(src)
    return src.new_zeros(size).scatter_reduce_(0, idx, src, "amax", include_self=False)


def remove_self_loops(edge_index: Tensor) -> Tensor:
    """``utils/loop.py: remove_self_loops`` (no edge_attr)."""
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask]


def add_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    """``utils/loop.py: add_self_loops`` (no edge_attr): append ``arange(N)`` loops."""
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)


def add_remaining_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    """``utils/loop.py: add_remaining_self_loops`` with ``edge_attr=None``:
    drop every existing self loop, append one loop per node."""
    mask = edge_index[0] != edge_index[1]
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index[:, mask], loop.unsqueeze(0).repeat(2, 1)], dim=1)


def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """``utils/_softmax.py: softmax(src, index, num_nodes=N)`` (ptr=None branch)."""
    src_max = scatter_max(src.detach(), index, num_nodes)
    out = src - src_max.index_select(0, index)
    out = out.exp()
    out_sum = scatter_sum(out, index, num_nodes) + 1e-16
    return out / out_sum.index_select(0, index)


# --------------------------------------------------------------------------- #
# nn/conv/gcn_conv.py: gcn_norm ; nn/conv/message_passing.py: propagate (sum)
# --------------------------------------------------------------------------- #
def gcn_norm(edge_index: Tensor, num_nodes: int, add_loops: bool,
             dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor]:
    """``gcn_norm(edge_index, None, N, improved=False, add_self_loops=add_loops,
    flow='source_to_target', dtype)`` (Tensor branch).  Returns the (possibly
    loop-augmented) ``edge_index`` and ``w_e = dis[row] * 1 * dis[col]`` with
    ``dis = deg^-1/2`` (inf -> 0) and ``deg`` the in-degree at ``col`` counting
    duplicate edges."""
    if add_loops:
        edge_index = add_remaining_self_loops(edge_index, num_nodes)
    edge_weight = torch.ones((edge_index.size(1),), dtype=dtype, device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    deg = scatter_sum(edge_weight, col, num_nodes)
    deg_inv_sqrt = deg.pow_(-0.5)
    deg_inv_sqrt.masked_fill_(deg_inv_sqrt == float("inf"), 0)
    edge_weight = deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]
    return edge_index, edge_weight


def propagate_sum(edge_index: Tensor, x: Tensor, edge_weight: Tensor) -> Tensor:
    """``MessagePassing.propagate`` for ``aggr='add'``, ``flow='source_to_target'``
    with ``message = edge_weight.view(-1,1) * x_j``: gather at ``edge_index[0]``,
    scale, ``scatter_add_`` at ``edge_index[1]``.  Materialises the ``[E,F]``
    temporaries exactly as PyG does."""
    x_j = x.index_select(0, edge_index[0])
    msg = edge_weight.view(-1, 1) * x_j
    return scatter_sum(msg, edge_index[1], x.size(0))


# --------------------------------------------------------------------------- #
# nn/dense/linear.py
# --------------------------------------------------------------------------- #
class Linear(nn.Module):
    """``torch_geometric.nn.dense.Linear``: ``weight [out,in]``, optional bias;
    ``weight_initializer`` None -> kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(in));
    'glorot' -> U(+-sqrt(6/(in+out)))."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True,
                 weight_initializer: Optional[str] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer = weight_initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == "glorot":
            a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        else:
            a = 1.0 / math.sqrt(self.in_channels)
        with torch.no_grad():
            self.weight.uniform_(-a, a)
            if self.bias is not None:
                self.bias.zero_()

    def forward(self, x: Tensor) -> Tensor:
        return F.linear(x, self.weight, self.bias)


def _glorot(t: Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


# --------------------------------------------------------------------------- #
# nn/conv/tag_conv.py
# --------------------------------------------------------------------------- #
class TAGConv(nn.Module):
    """``TAGConv(in, out, K=3, bias=True, normalize=True)``:
    ``out = sum_k lins[k](A_hat^k x) + bias``, ``A_hat`` from
    ``gcn_norm(add_self_loops=False)``."""

    def __init__(self, in_channels: int, out_channels: int, K: int = 3,
                 bias: bool = True, normalize: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.K, self.normalize = \
            in_channels, out_channels, K, normalize
        self.lins = nn.ModuleList(
            [Linear(in_channels, out_channels, bias=False) for _ in range(K + 1)])
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.size(0)
        if self.normalize:
            edge_index, w = gcn_norm(edge_index, n, add_loops=False, dtype=x.dtype)
        else:
            w = torch.ones(edge_index.size(1), dtype=x.dtype, device=x.device)
        out = self.lins[0](x)
        for lin in self.lins[1:]:
            x = propagate_sum(edge_index, x, w)
            out = out + lin(x)
        if self.bias is not None:
            out = out + self.bias
        return out


# --------------------------------------------------------------------------- #
# nn/conv/gcn_conv.py
# --------------------------------------------------------------------------- #
class GCNConv(nn.Module):
    """``GCNConv(in, out)`` defaults: ``add_self_loops=True, normalize=True,
    bias=True``; ``x' = lin(x)`` (glorot, no bias), one propagate, ``+ bias``."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        edge_index, w = gcn_norm(edge_index, x.size(0), add_loops=True, dtype=x.dtype)
        x = self.lin(x)
        out = propagate_sum(edge_index, x, w)
        if self.bias is not None:
            out = out + self.bias
        return out


# --------------------------------------------------------------------------- #
# nn/conv/gat_conv.py
# --------------------------------------------------------------------------- #
class GATConv(nn.Module):
    """``GATConv(in, out)`` defaults: ``heads=1, concat=True, negative_slope=0.2,
    dropout=0.0, add_self_loops=True, edge_dim=None, bias=True``.  PyG 2.5.x
    keeps a single ``lin`` when ``in_channels`` is an int (state_dict keys
    ``att_src, att_dst, bias, lin.weight``)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1,
                 negative_slope: float = 0.2, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.negative_slope = negative_slope
        self.lin = Linear(in_channels, heads * out_channels, bias=False,
                          weight_initializer="glorot")
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        else:
            self.register_parameter("bias", None)
        _glorot(self.att_src)
        _glorot(self.att_dst)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        H, C, n = self.heads, self.out_channels, x.size(0)
        h = self.lin(x).view(-1, H, C)
        alpha_src = (h * self.att_src).sum(dim=-1)          # [N,H]
        alpha_dst = (h * self.att_dst).sum(dim=-1)
        edge_index = add_self_loops(remove_self_loops(edge_index), n)
        row, col = edge_index[0], edge_index[1]
        # edge_update: alpha_j + alpha_i -> leaky_relu -> softmax over incoming edges of i
        alpha = alpha_src.index_select(0, row) + alpha_dst.index_select(0, col)
        alpha = F.leaky_relu(alpha, self.negative_slope)
        alpha = segment_softmax(alpha, col, n)              # [E,H]
        # message: alpha.unsqueeze(-1) * x_j ; aggregate: sum at col
        msg = alpha.unsqueeze(-1) * h.index_select(0, row)  # [E,H,C]
        out = scatter_sum(msg, col, n).view(-1, H * C)
        if self.bias is not None:
            out = out + self.bias
        return out


def knn(*args, **kwargs):  # imported but never called by models/model.py:2
    raise NotImplementedError("torch_geometric.nn.knn is dead code in the reference")


# --------------------------------------------------------------------------- #
# data/data.py, data/batch.py  (only the surface the reference touches:
# .x/.edge_index/.pos, clone(), to(), Batch.from_data_list, batch[i])
# --------------------------------------------------------------------------- #
class Data:
    def __init__(self, x=None, edge_index=None, pos=None, **kw):
        self.x, self.edge_index, self.pos = x, edge_index, pos
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        for t in (self.x, self.pos):
            if t is not None:
                return t.size(0)
        return int(self.edge_index.max()) + 1

    def _tensor_items(self):
        return [(k, v) for k, v in self.__dict__.items() if isinstance(v, Tensor)]

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        out.__dict__ = {k: (v.clone() if isinstance(v, Tensor) else v)
                        for k, v in self.__dict__.items()}
        return out

    def to(self, device):
        for k, v in self._tensor_items():
            setattr(self, k, v.to(device))
        return self


class Batch(Data):
    """``Batch.from_data_list``: cat ``x``/``pos`` on dim 0, ``edge_index`` on
    dim 1 with cumulative node offsets; ``batch`` [N] int64, ``ptr`` [B+1]."""

    @classmethod
    def from_data_list(cls, data_list):
        counts = [d.num_nodes for d in data_list]
        ptr = torch.tensor([0] + counts, dtype=torch.long).cumsum(0)
        out = cls()
        out.x = torch.cat([d.x for d in data_list], 0) if data_list[0].x is not None else None
        out.pos = torch.cat([d.pos for d in data_list], 0) if data_list[0].pos is not None else None
        out.edge_index = torch.cat(
            [d.edge_index + int(ptr[i]) for i, d in enumerate(data_list)], dim=1)
        out.batch = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts))
        out.ptr = ptr
        out._edge_ptr = torch.tensor(
            [0] + [d.edge_index.size(1) for d in data_list], dtype=torch.long).cumsum(0)
        return out

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel()) - 1

    def __getitem__(self, i: int) -> Data:
        a, b = int(self.ptr[i]), int(self.ptr[i + 1])
        ea, eb = int(self._edge_ptr[i]), int(self._edge_ptr[i + 1])
        return Data(x=None if self.x is None else self.x[a:b],
                    pos=None if self.pos is None else self.pos[a:b],
                    edge_index=self.edge_index[:, ea:eb] - a)
