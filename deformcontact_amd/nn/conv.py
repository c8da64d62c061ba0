"""``TAGConv`` / ``GCNConv`` / ``GATConv`` on the HIP hop kernels.

Drop-in for the PyG classes the reference instantiates at
``/root/reference/models/model.py:39-50`` and calls at ``:71,77``.  Parameter
names, shapes and default initialisers follow PyG 2.5.2 so reference
checkpoints (``eval.py:36,89``: ``load_state_dict(torch.load(...))``) load
unchanged:

* ``TAGConv``: ``lins.{0..K}.weight [out,in]`` (no per-lin bias), ``bias [out]``;
* ``GCNConv``: ``lin.weight [out,in]`` (glorot), ``bias [out]``;
* ``GATConv``: ``lin.weight [H*out,in]`` (glorot), ``att_src/att_dst [1,H,out]``,
  ``bias [H*out]``.

No CPU path: calling a conv with CPU tensors raises.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..graph import GraphIndex, _require_cuda, graph_index
from ..deferred import deferred, resolve

#: a plain ``conv(x, edge_index)`` call returns a ``deferred.DeferredActivation``: the ``F.relu`` the reference applies
#: right behind it (``models/model.py:71,77``) then runs fused in the layer's epilogue, and the output lands in the hop
#: slab of the TAGConv layer that consumed it last time (``TAGConv._consumer_geom``) - the unchanged reference wiring on
#: the same launches as ``graphnet.ContactEncoder``.  False / ``DC_DEFER_ACT=0``: the conv returns its output directly.
DEFER_ACTIVATION = os.environ.get("DC_DEFER_ACT", "1") != "0"


#: OPT-IN (``DC_BRANCH_STREAMS=1`` or ``nn.conv.BRANCH_STREAMS = True``; default off): run the independent branches of a
#: forward pass written with plain conv calls - the resting loop and the rigid loop of ``models/model.py:69-78`` - on two HIP
#: streams, as ``graphnet.ContactEncoder`` does.  A conv whose input carries no producer tag starts a branch: the first
#: branch of a pass stays on the caller's stream, every further one runs on a side stream that waits only for the point at
#: which the pass began (an event recorded at the pass's first conv call), and the caller's stream waits for the side stream
#: behind every layer launched there, so whatever the caller enqueues later (the cross-attention) sees the results; autograd
#: replays the structure in backward.  CONTRACT, which is why this is opt-in: the inputs of EVERY branch (features and
#: ``edge_index`` of both graphs) must be complete on the caller's stream when the first conv of the pass is called - true of
#: a model whose ``forward`` receives its graphs as arguments, as the reference's does.  A pass ends when a branch's first
#: layer is called again.
BRANCH_STREAMS = os.environ.get("DC_BRANCH_STREAMS", "0") == "1"
_PASS = {}            # device index -> {"roots": set of module ids, "event", "main": stream handle, "count"}


def _branch_stream(module: nn.Module, x: Tensor):
    """The side stream this plain conv call's layer runs on, or None (the caller's stream).  See ``BRANCH_STREAMS``."""
    if not BRANCH_STREAMS:
        return None
    side = getattr(x, "_dc_branch", None)
    if side is not None:
        return side                                            # a later layer of a side branch follows its input
    if getattr(x, "_dc_producer", None) is not None or x.grad_fn is not None:
        return None                                            # not a graph input: ordinary stream semantics
    from ..graphnet import ContactEncoder
    dev = x.device
    main = torch.cuda.current_stream(dev)
    st = _PASS.get(dev.index)
    if st is None or id(module) in st["roots"] or st["main"] != main.cuda_stream:
        st = _PASS[dev.index] = {"roots": set(), "event": torch.cuda.Event(), "main": main.cuda_stream, "count": 0}
        st["event"].record(main)                               # the pass begins: every branch's inputs exist (contract)
    st["roots"].add(id(module))
    st["count"] += 1
    if st["count"] == 1:
        return None
    side = ContactEncoder._side_stream(dev)
    side.wait_event(st["event"])
    return side


def _on_branch(side, fn):
    """``fn()`` on ``side`` (None: here); the caller's stream then waits for it and the result remembers its stream."""
    if side is None:
        return fn()
    main = torch.cuda.current_stream(side.device)
    with torch.cuda.stream(side):
        out = fn()
    main.wait_stream(side)
    out.record_stream(main)
    out._dc_branch = side
    return out


def _grad_wanted(x: Tensor, module: nn.Module) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


class _Lin(nn.Module):
    """Parameter holder mirroring ``torch_geometric.nn.dense.Linear(bias=False)``."""

    #: set by GATConv on its ``lin``: keep the six-product dense kernels (``ops.dense_linear(six_products=True)``)
    six_products = False

    def __init__(self, in_channels: int, out_channels: int, initializer: Optional[str] = None):
        super().__init__()
        self.in_channels, self.out_channels, self.initializer = in_channels, out_channels, initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        if self.initializer == "glorot":
            a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        else:  # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
            a = 1.0 / math.sqrt(self.in_channels)
        with torch.no_grad():
            self.weight.uniform_(-a, a)

    def forward(self, x: Tensor) -> Tensor:
        x = resolve(x)
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            return ops.dense_linear(x, self.weight, six_products=self.six_products)      # the library's MFMA dense block
        return torch.nn.functional.linear(x, self.weight)


def _check_inputs(x: Tensor, edge_index: Tensor, in_channels: int):
    _require_cuda(x, "x")
    _require_cuda(edge_index, "edge_index")
    if x.dim() != 2 or x.size(1) != in_channels:
        raise ValueError(f"x must be [N, {in_channels}], got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise ValueError(f"x must be float32, got {x.dtype}")
    if x.device != edge_index.device:
        raise RuntimeError(f"x is on {x.device} but edge_index is on {edge_index.device}")


class TAGConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, K: int = 3, bias: bool = True,
                 normalize: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.K, self.normalize = \
            in_channels, out_channels, K, normalize
        self.lins = nn.ModuleList([_Lin(in_channels, out_channels) for _ in range(K + 1)])
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        #: {activation fused?: (width, padded width) of the hop slab of the layer that consumed the last output}
        self._consumer_geom = {}

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def graph_flags(self) -> dict:
        return dict(self_loops=False, normalize=self.normalize)

    def graph(self, edge_index: Tensor, num_nodes: int, segments=None) -> GraphIndex:
        return graph_index(edge_index, num_nodes, segments=segments, **self.graph_flags())

    supports_fused_relu = True
    #: dtype of the output when the input is bfloat16 (the bf16-storage forward path)
    bf16_out = torch.bfloat16

    def slab_width(self) -> int:
        return ops.tag_slab_geometry(self.in_channels, self.K)[2]

    def forward(self, x: Tensor, edge_index: Tensor, relu: bool = False,
                next_conv: "Optional[TAGConv]" = None, out_into: Optional[Tensor] = None) -> Tensor:
        """``conv(x, edge_index)`` as PyG.  Extensions: ``relu=True`` fuses the ReLU the reference
        applies right after (``models/model.py:71,77``) into the MFMA epilogue; ``next_conv`` (the
        TAGConv that consumes this output) lets the output be written straight into that
        layer's hop slab; ``out_into`` (a ``[N, out]`` row-major view the caller owns, e.g. this
        branch's rows of a merged slab - ``ops.merged_slab_part``) receives the output instead."""
        x = resolve(x)                 # (the deferred result of another plain conv call: its value, see deferred.py)
        if x.dtype == torch.bfloat16:
            # bf16-STORED features (BASELINE.json configs[4]): bf16 hops with fp32 accumulation + the
            # bf16 MFMA dense block; forward only.  ``out_dtype`` of the last layer: self.bf16_out
            _require_cuda(x, "x")
            _require_cuda(edge_index, "edge_index")
            if x.dim() != 2 or x.size(1) != self.in_channels:
                raise ValueError(f"x must be [N, {self.in_channels}], got {tuple(x.shape)}")
            g = self.graph(edge_index, x.size(0))
            nk = next_conv.K if (isinstance(next_conv, TAGConv)
                                 and next_conv.in_channels == self.out_channels) else None
            return ops.tag_conv_bf16(g, x, [lin.weight for lin in self.lins], self.bias, relu=relu,
                                     out_dtype=self.bf16_out, next_k=nk)
        _check_inputs(x, edge_index, self.in_channels)
        self._note_consumer(x)
        if DEFER_ACTIVATION and not relu and next_conv is None and out_into is None:
            # the PyG call as the reference makes it: what follows decides (deferred.py) - F.relu runs fused
            side = _branch_stream(self, x)

            def run(act: bool) -> Tensor:
                def layer():
                    g = self.graph(edge_index, x.size(0))
                    return self._mark(ops.tag_conv(g, x, [lin.weight for lin in self.lins], self.bias, relu=act,
                                                   next_geom=self._consumer_geom.get(act)), act)
                return _on_branch(side, layer)
            if side is None:
                self.graph(edge_index, x.size(0))        # the adjacency build starts at the call, as without deferral
            return deferred(run, x.size(0), self.out_channels, x, _grad_wanted(x, self)).guard(
                x, edge_index, *self.parameters())
        g = self.graph(edge_index, x.size(0))
        nxt = None
        if out_into is not None:
            nxt = ops.OutInto(out_into)
        elif isinstance(next_conv, TAGConv) and next_conv.in_channels == self.out_channels:
            nxt = ops.tag_slab_geometry(next_conv.in_channels, next_conv.K)[1:]
        return self._mark(ops.tag_conv(g, x, [lin.weight for lin in self.lins], self.bias, relu=relu,
                                       next_geom=nxt), relu)

    # ---- the consumer of a layer's output, discovered at call time ------------------------------------------------
    # ``models/model.py:69-78`` calls the layers one by one; nothing tells a conv who reads its output.  Every output
    # carries a weak reference to the layer that produced it; a TAGConv that is handed such a tensor and does NOT find
    # it sitting in block 0 of a hop slab of its own geometry tells the producer, which from its next call on
    # allocates its output as block 0 of that slab (one step of a training loop runs with the packing copy, every later
    # one without).  A wrong guess costs memory, never correctness: the output is a view either way.
    def _mark(self, out: Tensor, act: bool) -> Tensor:
        out._dc_producer = (weakref.ref(self), bool(act))
        return out

    def _note_consumer(self, x: Tensor) -> None:
        tag = getattr(x, "_dc_producer", None)
        if tag is None:
            return
        prod, act = tag[0](), tag[1]
        if isinstance(prod, TAGConv) and prod is not self and prod.out_channels == self.in_channels:
            geom = ops.tag_slab_geometry(self.in_channels, self.K)[1:]
            if prod._consumer_geom.get(act) != geom:
                prod._consumer_geom[act] = geom

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, K={self.K}"


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = _Lin(in_channels, out_channels, initializer="glorot")
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def graph_flags(self) -> dict:
        return dict(self_loops=True, normalize=True)

    def graph(self, edge_index: Tensor, num_nodes: int, segments=None) -> GraphIndex:
        return graph_index(edge_index, num_nodes, segments=segments, **self.graph_flags())

    supports_fused_relu = True

    def forward(self, x: Tensor, edge_index: Tensor, relu: bool = False, next_conv=None) -> Tensor:
        """``conv(x, edge_index)`` as PyG; ``relu=True`` fuses the ReLU the reference applies right after
        (``models/model.py:71,77``) - with the bias - into the aggregation launch (``ops.gcn_aggregate``)."""
        x = resolve(x)
        _check_inputs(x, edge_index, self.in_channels)
        if DEFER_ACTIVATION and not relu and next_conv is None:
            side = _branch_stream(self, x)
            return deferred(lambda act: _on_branch(side, lambda: self._layer(self.graph(edge_index, x.size(0)), x, act)),
                            x.size(0), self.out_channels, x, _grad_wanted(x, self)).guard(x, edge_index, *self.parameters())
        g = self.graph(edge_index, x.size(0))
        return self._layer(g, x, relu)

    def _layer(self, g: GraphIndex, x: Tensor, relu: bool) -> Tensor:
        h = self.lin(x)
        if ops.fused_gnn_ok(h):
            return ops.gcn_aggregate(g, h, self.bias, relu)
        out = ops.propagate(g, h, weighted=True)
        if self.bias is not None:
            out = out + self.bias
        return torch.relu(out) if relu else out


class GATConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, heads: int = 1,
                 negative_slope: float = 0.2, bias: bool = True):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("the reference only uses heads=1 (models/model.py:45,49)")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.negative_slope = negative_slope
        self.lin = _Lin(in_channels, heads * out_channels, initializer="glorot")
        self.lin.six_products = True             # the attention vectors' gradient cancels to 1 % of its terms: 24-bit products
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        a = math.sqrt(6.0 / (self.heads + self.out_channels))
        with torch.no_grad():
            self.att_src.uniform_(-a, a)
            self.att_dst.uniform_(-a, a)
            if self.bias is not None:
                self.bias.zero_()

    def graph_flags(self) -> dict:
        return dict(self_loops=True, normalize=False)

    def graph(self, edge_index: Tensor, num_nodes: int, segments=None) -> GraphIndex:
        return graph_index(edge_index, num_nodes, segments=segments, **self.graph_flags())

    supports_fused_relu = True

    def forward(self, x: Tensor, edge_index: Tensor, relu: bool = False, next_conv=None) -> Tensor:
        """``conv(x, edge_index)`` as PyG; everything behind ``lin`` is one autograd node on fused kernels
        (``ops.gat_conv``); ``relu=True`` also fuses the encoder's ReLU (``models/model.py:71,77``)."""
        x = resolve(x)
        _check_inputs(x, edge_index, self.in_channels)
        if DEFER_ACTIVATION and not relu and next_conv is None:
            side = _branch_stream(self, x)
            return deferred(lambda act: _on_branch(side, lambda: self._layer(self.graph(edge_index, x.size(0)), x, act)),
                            x.size(0), self.heads * self.out_channels, x, _grad_wanted(x, self)).guard(
                                x, edge_index, *self.parameters())
        g = self.graph(edge_index, x.size(0))
        return self._layer(g, x, relu)

    def _layer(self, g: GraphIndex, x: Tensor, relu: bool) -> Tensor:
        h = self.lin(x)
        if ops.fused_gnn_ok(h):
            return ops.gat_conv(g, h, self.att_src, self.att_dst, self.bias, self.negative_slope, relu)
        a_src = (h * self.att_src.view(1, -1)).sum(-1)
        a_dst = (h * self.att_dst.view(1, -1)).sum(-1)
        out = ops.gat_aggregate(g, h, a_src, a_dst, self.negative_slope)
        if self.bias is not None:
            out = out + self.bias
        return torch.relu(out) if relu else out


def knn(*args, **kwargs):
    """Imported by ``models/model.py:2`` but never called by the reference."""
    raise NotImplementedError("torch_geometric.nn.knn is dead code in the reference")
