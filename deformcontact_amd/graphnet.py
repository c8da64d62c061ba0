"""Encoder / full-network harness with the reference's parameter names.

``ContactEncoder`` is the hot path as the reference wires it
(``/root/reference/models/model.py:39-50`` construction, ``:69-78`` forward):
two branches (soft "resting" mesh, rigid contact sphere) of ``encoder_layers``
conv layers, each followed by ReLU and dropout.  ``GraphNet`` adds the parts that
are out of the hot path but needed for full-step parity (``:7-21`` unmasked
multi-head cross attention, ``:52-64`` decoder, ``:91-95`` residual output) on
stock PyTorch.  ``state_dict()`` keys equal the reference's, so its checkpoints
load (``eval.py:36,89``).

The reference's own ``models/model.py`` also runs unchanged on this package via
``deformcontact_amd.install_as_torch_geometric()``; this module exists so that
bench / tests / training on the GPU box need no file from ``/root/reference``.
"""
from __future__ import annotations

import os

from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn as dc_nn
from . import ops


def _conv_class(backbone: str, conv_module):
    mod = conv_module if conv_module is not None else dc_nn
    # same fall-through as the reference: anything that is not GAT/GCN is TAG
    name = backbone if backbone in ("GATConv", "GCNConv") else "TAGConv"
    return getattr(mod, name)


def _segments_of(batch):
    """Host-side layout of a ``data.Batch`` (None for anything else, or once its ``edge_index`` was replaced)."""
    f = getattr(batch, "segments", None)
    return f() if callable(f) else None


class ContactEncoder(nn.Module):
    def __init__(self, input_dims: Sequence[int], hidden_dim: int, encoder_layers: int = 2,
                 dropout_rate: float = 0.0, backbone: str = "TAGConv", conv_module=None):
        super().__init__()
        conv = _conv_class(backbone, conv_module)
        self.backbone, self.dropout_rate, self.encoder_layers = backbone, dropout_rate, encoder_layers
        widths_rest = [input_dims[0]] + [hidden_dim] * encoder_layers
        widths_rig = [input_dims[1]] + [hidden_dim] * encoder_layers
        self.conv_layers_resting = nn.ModuleList(
            conv(a, b) for a, b in zip(widths_rest[:-1], widths_rest[1:]))
        self.conv_layers_rigid = nn.ModuleList(
            conv(a, b) for a, b in zip(widths_rig[:-1], widths_rig[1:]))

    def _branch(self, layers, x, edge_index, segments=None):
        drop = self.dropout_rate > 0.0 and self.training
        if segments is not None and hasattr(layers[0], "graph") and x.is_cuda:
            # the batch layout is known: the sorted adjacency every layer will look up comes from the one-launch
            # segmented build (graph.graph_index(segments=)) instead of the 5-launch global pipeline
            layers[0].graph(edge_index, x.size(0), segments=segments)
        for i, conv in enumerate(layers):
            if getattr(conv, "supports_fused_relu", False):
                # ReLU in the MFMA epilogue; output written into the next layer's hop slab
                nxt = layers[i + 1] if (i + 1 < len(layers) and not drop) else None
                x = conv(x, edge_index, relu=True, next_conv=nxt)
            else:
                x = F.relu(conv(x, edge_index))
            if drop:
                x = F.dropout(x, p=self.dropout_rate, training=True)
        return x

    #: run the soft and the rigid branch on two HIP streams (they are independent until the
    #: cross-attention): each branch's kernels fill the tail of the other's, and the short hop
    #: kernels of one branch hide under the dense blocks of the other.  Autograd replays the
    #: same two-stream structure in backward; inside hipGraph capture it becomes two parallel
    #: graph branches.
    overlap_branches = True
    _side_streams = {}

    @classmethod
    def _side_stream(cls, device) -> "torch.cuda.Stream":
        key = (device.type, device.index)
        if key not in cls._side_streams:
            cls._side_streams[key] = torch.cuda.Stream(device=device)
            ops.DW_LAST_STREAMS.add(cls._side_streams[key].cuda_stream)      # (ops.DW_LAST_STREAMS: why)
        return cls._side_streams[key]

    #: run both branches as ONE block-diagonal problem: one sorted adjacency over the
    #: merged node space (``graph.merged_graph_index``), every layer after the first as single launches for
    #: both branches (``ops.tag_conv_grouped``: merged hops, grouped dense blocks - same rows, same
    #: arithmetic, bit-identical outputs and gradients), the narrow first layers per branch over row windows
    #: of the merged adjacency, writing into the merged slab.  Applies to TAGConv encoders whose later
    #: layers are 256 wide (the shipped config); ``DC_MERGE_BRANCHES=0`` keeps one launch per branch.
    merge_branches = os.environ.get("DC_MERGE_BRANCHES", "0") == "1"

    def _mergeable(self, x_s, x_r) -> bool:
        if not (self.merge_branches and x_s.is_cuda and x_r.is_cuda and x_s.dtype == torch.float32
                and x_r.dtype == torch.float32 and x_s.size(0) > 0 and x_r.size(0) > 0):
            return False
        if self.dropout_rate > 0.0 and self.training:
            return False
        ls, lr = self.conv_layers_resting, self.conv_layers_rigid
        if len(ls) < 2 or len(ls) != len(lr):
            return False
        flags = None
        for i, (a, b) in enumerate(zip(ls, lr)):
            if type(a) is not dc_nn.TAGConv or type(b) is not dc_nn.TAGConv:
                return False
            for c in (a, b):
                if flags is None:
                    flags = c.graph_flags()
                if c.graph_flags() != flags:
                    return False
            if i == 0:
                continue                  # first layers run per branch over row windows of the merged adjacency
            if not ((a.in_channels, a.out_channels, a.K) == (b.in_channels, b.out_channels, b.K)
                      and (a.bias is None) == (b.bias is None)
                      and ops.grouped_eligible(a.in_channels, a.out_channels, a.K)):
                return False
        return ls[0].out_channels == ls[1].in_channels and lr[0].out_channels == lr[1].in_channels

    def topology(self, graph_resting, graph_rigid):
        """The sorted adjacency object(s) ``encode`` will use for these batches (built if needed): the
        merged one, or one per branch.  For callers that prepare the topology ahead of the step or
        mark it constant (``_static_ok``) for a captured graph."""
        x_s, e_s, x_r, e_r = graph_resting.x, graph_resting.edge_index, graph_rigid.x, graph_rigid.edge_index
        if self._mergeable(x_s, x_r):
            from .graph import merged_graph_index
            return [merged_graph_index([(e_s, x_s.size(0)), (e_r, x_r.size(0))],
                                       **self.conv_layers_resting[0].graph_flags())]
        return [c.graph(e, x.size(0), segments=_segments_of(b))
                for c, e, x, b in ((self.conv_layers_resting[0], e_s, x_s, graph_resting),
                                   (self.conv_layers_rigid[0], e_r, x_r, graph_rigid))
                if hasattr(c, "graph")]

    def _encode_merged(self, x_s, e_s, x_r, e_r):
        from .graph import merged_graph_index
        ls, lr = self.conv_layers_resting, self.conv_layers_rigid
        mg = merged_graph_index([(e_s, x_s.size(0)), (e_r, x_r.size(0))], **ls[0].graph_flags())
        slab = ops.alloc_merged_slab(mg, ls[1].in_channels, ls[1].K, x_s.device)
        fi = ls[1].in_channels
        into_s, into_r = ops.merged_slab_part(slab, mg, 0, fi), ops.merged_slab_part(slab, mg, 1, fi)
        if self.overlap_branches:
            # the two narrow first layers side by side (each has its own small launches)
            main = torch.cuda.current_stream(x_s.device)
            side = self._side_stream(x_s.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                h_r = lr[0](x_r, e_r, relu=True, out_into=into_r)
            h_s = ls[0](x_s, e_s, relu=True, out_into=into_s)
            main.wait_stream(side)
            h_r.record_stream(main)
        else:
            h_s = ls[0](x_s, e_s, relu=True, out_into=into_s)
            h_r = lr[0](x_r, e_r, relu=True, out_into=into_r)
        xs = (h_s, h_r)
        for i in range(1, len(ls)):
            nk = ls[i + 1].K if i + 1 < len(ls) else None
            xs = ops.tag_conv_grouped(mg, xs, [[lin.weight for lin in c.lins] for c in (ls[i], lr[i])],
                                      [ls[i].bias, lr[i].bias], relu=True, next_k=nk)
        return xs[0], xs[1]

    def encode(self, graph_resting, graph_rigid) -> Tuple[torch.Tensor, torch.Tensor]:
        x_s, e_s = graph_resting.x, graph_resting.edge_index
        x_r, e_r = graph_rigid.x, graph_rigid.edge_index
        if self._mergeable(x_s, x_r):
            return self._encode_merged(x_s, e_s, x_r, e_r)
        seg_s, seg_r = _segments_of(graph_resting), _segments_of(graph_rigid)
        if not (self.overlap_branches and x_s.is_cuda and x_r.is_cuda):
            return (self._branch(self.conv_layers_resting, x_s, e_s, seg_s),
                    self._branch(self.conv_layers_rigid, x_r, e_r, seg_r))
        # each branch builds (or finds) its own sorted adjacency on its own stream: the two
        # dc_graph_build pipelines of a new batch run side by side; every later use of the rigid
        # one on the caller's stream is ordered behind the join below
        main = torch.cuda.current_stream(x_s.device)
        side = self._side_stream(x_s.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out_r = self._branch(self.conv_layers_rigid, x_r, e_r, seg_r)
        out_s = self._branch(self.conv_layers_resting, x_s, e_s, seg_s)
        main.wait_stream(side)
        out_r.record_stream(main)
        return out_s, out_r

    def forward(self, graph_resting, graph_rigid):
        return self.encode(graph_resting, graph_rigid)


class ReferenceWiring(ContactEncoder):
    """The two encoder loops written the way the reference writes them (``/root/reference/models/model.py:69-78``):
    plain ``conv(x, graph.edge_index)`` calls, ``F.relu`` and ``F.dropout`` outside the conv, one branch after the other
    on the caller's stream, nothing passed along but the ``edge_index`` tensor of a ``Batch.from_data_list(...).to(dev)``
    batch (``train.py:36-46``).  Same parameters as ``ContactEncoder``.  This is what the reference's own ``GraphNet`` does
    through ``install_as_torch_geometric()``; it exists so that ``bench.py`` and the GPU tests can run that wiring where
    ``/root/reference`` is absent.  The layout travels on the ``edge_index`` tensor (``data.Batch._tag_edge_layout``), the
    ReLU is fused by ``deferred``, the next layer's slab is found by ``TAGConv._note_consumer``: the launches are those
    of ``ContactEncoder`` on one stream."""

    def encode(self, graph_resting, graph_rigid):
        x_resting = graph_resting.x
        for conv in self.conv_layers_resting:
            x_resting = F.relu(conv(x_resting, graph_resting.edge_index))
            x_resting = F.dropout(x_resting, p=self.dropout_rate, training=self.training)
        x_rigid = graph_rigid.x
        for conv in self.conv_layers_rigid:
            x_rigid = F.relu(conv(x_rigid, graph_rigid.edge_index))
            x_rigid = F.dropout(x_rigid, p=self.dropout_rate, training=self.training)
        return x_resting, x_rigid


class CrossAttention(nn.Module):
    """The reference's ``MultiHeadAttention``: per head one ``Linear(d, d)`` shared by
    queries and keys, raw rigid features as values, no 1/sqrt(d), no per-graph mask."""

    def __init__(self, feature_dim: int, num_heads: int):
        super().__init__()
        self.attention_heads = nn.ModuleList(
            nn.Linear(feature_dim, feature_dim) for _ in range(num_heads))

    #: the library's attention (``attention.attention_core``: at d = 256 the flash-style forward ``dc_attn_flash_fwd``,
    #: scores never leave the compute unit, and the ``dc_attn_flash_ds`` backward; other widths the blocked form) -
    #: the ``[N_s, N_r]`` score / weight matrices of the reference are not materialised in the forward.
    #: "auto" (default) uses it once a score matrix would exceed ``fused_min_scores`` elements (batch 32: 8e8 scores,
    #: whole step 24 - 25 ms against 51 ms / 15.6 GiB with the materialising formula on stock PyTorch, DESIGN.md 4.6;
    #: below the threshold - the shipped batch 4 - the materialising formula has fewer launches and is faster);
    #: ``DC_FUSED_ATTN=1`` / ``0`` force it on / off.
    fused = os.environ.get("DC_FUSED_ATTN", "auto")
    fused_min_scores = 1 << 26
    #: rows (soft + rigid) up to which the heads' shared Linear runs once over both graphs' rows (see ``forward``)
    joint_max_rows = int(os.environ.get("DC_ATTN_JOINT_ROWS", "16384"))
    #: with the joint Linear: all heads through one dense block and batched score / pooling products (stock attention path)
    batched_heads = os.environ.get("DC_ATTN_BATCHED_HEADS", "1") != "0"

    #: SURVEY.md 8(f) rank 1, third part - an OPTION the reference does not have and that is OFF by default (parity): the
    #: reference's attention is unmasked across the batch (``models/model.py:16-18``: a soft node attends to the rigid
    #: nodes of EVERY sample of the device batch), which makes a sample's prediction depend on which other
    #: samples share its batch and costs B x the work.  ``per_graph_mask = True`` restricts every soft node to the rigid
    #: nodes of its own sample (a block-diagonal mask over the batch layout ``Batch.from_data_list`` records): the
    #: batch-size-independent form, B separate [N_s/B, N_r/B] attentions instead of one [N_s, N_r] one.
    per_graph_mask = False

    def _per_graph(self, x_resting, x_rigid, seg_rest, seg_rig):
        ns, nr = [int(v) for v in seg_rest[0]], [int(v) for v in seg_rig[0]]
        if len(ns) != len(nr) or ns[-1] != x_resting.size(0) or nr[-1] != x_rigid.size(0):
            raise ValueError("per-graph attention mask: the two batches must hold the same number of graphs and their "
                             "layouts must cover the feature matrices")
        b = len(ns) - 1
        ds, dr = {ns[i + 1] - ns[i] for i in range(b)}, {nr[i + 1] - nr[i] for i in range(b)}
        pooled = []
        for head in self.attention_heads:
            q, k = _linear(head, x_resting), _linear(head, x_rigid)
            if len(ds) == 1 and len(dr) == 1 and 0 not in dr:
                # equal-size graphs (the everyday shape): one batched product per step
                qs, ks, vs = q.view(b, -1, q.size(1)), k.view(b, -1, k.size(1)), x_rigid.view(b, -1, x_rigid.size(1))
                pooled.append((torch.softmax(qs @ ks.transpose(1, 2), dim=-1) @ vs).reshape(x_resting.size(0), -1))
            else:
                parts = []
                for i in range(b):
                    qi, ki, vi = q[ns[i]:ns[i + 1]], k[nr[i]:nr[i + 1]], x_rigid[nr[i]:nr[i + 1]]
                    if ki.size(0) == 0:                  # a sample without rigid nodes: nothing to attend to
                        parts.append(qi.new_zeros(qi.size(0), x_rigid.size(1)))
                    else:
                        parts.append(torch.softmax(qi @ ki.t(), dim=-1) @ vi)
                pooled.append(torch.cat(parts, dim=0))
        return torch.cat(pooled, dim=-1)

    def forward(self, x_resting, x_rigid, segments=None, return_list=False):
        if self.per_graph_mask:
            if segments is None or segments[0] is None or segments[1] is None:
                raise ValueError("per_graph_mask needs the layouts of both batches (data.Batch.segments())")
            out = self._per_graph(x_resting, x_rigid, segments[0], segments[1])
            return list(out.split(x_rigid.size(1), dim=-1)) if return_list else out
        pooled = []
        eligible = (x_resting.is_cuda and x_resting.dtype == torch.float32
                    and x_resting.size(1) % 16 == 0 and x_rigid.size(0) > 0)
        mode = str(self.fused).lower()
        use_fused = eligible and (mode in ("1", "true", "on") or (
            mode == "auto" and x_resting.size(0) * x_rigid.size(0) >= self.fused_min_scores))
        # queries and keys of a head share ONE Linear (models/model.py:11,16): one dense block over the rows of both graphs
        # instead of two - and one dX / dW block in backward instead of two plus an add of the two weight gradients.  Small
        # batches (the shipped batch 4) are bound by their launch count: 2 + 4 + 4 launches fewer per step; from
        # `joint_max_rows` rows on the copy that joins the rows costs more than the launches it saves.
        joint = (x_resting.is_cuda and x_resting.dtype == torch.float32 and x_rigid.dtype == torch.float32
                 and x_resting.dim() == 2 and x_resting.size(1) == x_rigid.size(1)
                 and x_resting.size(0) + x_rigid.size(0) <= self.joint_max_rows)
        xcat = torch.cat([x_resting, x_rigid], dim=0) if joint else None
        ns = x_resting.size(0)
        heads = list(self.attention_heads)
        if (joint and not use_fused and self.batched_heads and len(heads) > 1 and x_rigid.size(0) > 0
                and all(h.weight.shape == heads[0].weight.shape and (h.bias is None) == (heads[0].bias is None) for h in heads)):
            # ... and all heads at once: ONE dense block with the heads' weights stacked along the output ([N, H d]), the
            # scores / softmax / pooling of all heads as batched products over strided views of it - the same sums per head,
            # a third of the launches (the shipped batch 4 is bound by its launch count)
            hn, d = len(heads), heads[0].weight.size(0)
            w_cat = torch.cat([h.weight for h in heads], dim=0)
            b_cat = torch.cat([h.bias for h in heads], dim=0) if heads[0].bias is not None else None
            qk = ops.dense_linear(xcat, w_cat, b_cat)                          # [ns + nr, H d]
            qk3 = qk.view(qk.size(0), hn, d).transpose(0, 1)                   # [H, ns + nr, d], no copy
            p = torch.softmax(torch.bmm(qk3[:, :ns], qk3[:, ns:].transpose(1, 2)), dim=-1)
            o = torch.bmm(p, x_rigid.unsqueeze(0).expand(hn, -1, -1))          # [H, ns, dv]
            pooled = list(o.unbind(0))
            return pooled if return_list else torch.cat(pooled, dim=-1)
        for head in heads:
            if joint:
                qk = _linear(head, xcat)
                q, k = qk[:ns], qk[ns:]
            else:
                q, k = _linear(head, x_resting), _linear(head, x_rigid)
            if use_fused:
                from .attention import attention_core
                pooled.append(attention_core(q, k, x_rigid))
            else:
                pooled.append(torch.softmax(q @ k.t(), dim=-1) @ x_rigid)
        if return_list:
            return pooled
        return torch.cat(pooled, dim=-1)


def _linear(lin: nn.Linear, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """``act(lin(x))`` on the library's dense block (bias + ReLU in the MFMA epilogue, dX / dW on the
    same kernels) for float32 CUDA inputs; the parameters stay those of the ``nn.Linear``."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        return ops.dense_linear(x, lin.weight, lin.bias, relu=relu)
    y = lin(x)
    return F.relu(y) if relu else y


class GraphNet(ContactEncoder):
    def __init__(self, input_dims, hidden_dim, output_dim, encoder_layers, decoder_layers,
                 dropout_rate, knn_k=None, backbone="TAGConv", use_mha=True, num_mha_heads=2,
                 mode="res", conv_module=None):
        super().__init__(input_dims, hidden_dim, encoder_layers, dropout_rate, backbone, conv_module)
        self.mode, self.use_mha = mode, use_mha
        width = hidden_dim * (num_mha_heads + 1) if use_mha else hidden_dim * 2
        blocks = []
        for _ in range(decoder_layers):
            blocks += [nn.Linear(width, hidden_dim), nn.ReLU(), nn.Dropout(dropout_rate)]
            width = hidden_dim
        blocks.append(nn.Linear(hidden_dim, output_dim))
        self.decoder = nn.Sequential(*blocks)
        self.multihead_attention = CrossAttention(hidden_dim, num_mha_heads)

    def forward(self, graph_resting, graph_rigid):
        x_rest, x_rig = self.encode(graph_resting, graph_rigid)
        if self.multihead_attention.per_graph_mask:          # (opt-in; the reference attends across the whole batch)
            pooled = self.multihead_attention(x_rest, x_rig, (_segments_of(graph_resting), _segments_of(graph_rigid)),
                                              return_list=True)
        else:
            pooled = self.multihead_attention(x_rest, x_rig, return_list=True)     # always applied (reference quirk)
        delta = self._decode(torch.cat([x_rest] + list(pooled), dim=-1))   # one concatenation for [x | head 0 | head 1 ...]
        out = graph_resting.clone()
        if self.mode == "res":
            out.pos = out.pos + delta
        elif self.mode == "rec":
            out.pos = delta
        return out


def _decode_impl(decoder: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """``self.decoder(x)`` (``models/model.py:52-64``: Linear -> ReLU -> Dropout blocks + Linear) with
    every Linear (+ its ReLU) as one launch of the library's dense block."""
    mods = list(decoder)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = _linear(m, x, relu=fuse)
            i += 2 if fuse else 1
        else:
            x = m(x)
            i += 1
    return x


GraphNet._decode = lambda self, x: _decode_impl(self.decoder, x)


def gradient_consistency_loss(pred, target) -> torch.Tensor:
    """``models/losses.py:7-19``: mean over edges of the L2 norm of the difference of
    edge vectors between prediction and target."""
    ei_t, ei_p = target.edge_index, pred.edge_index
    d_t = target.pos[ei_t[1]] - target.pos[ei_t[0]]
    d_p = pred.pos[ei_p[1]] - pred.pos[ei_p[0]]
    return (d_t - d_p).norm(p=2, dim=-1).sum() / ei_t.size(1)


def load_model(network_cfg, conv_module=None) -> GraphNet:
    """``models/model_loader.py:3-16`` for a ``config.network``-like object or dict."""
    get = (lambda k: network_cfg[k]) if isinstance(network_cfg, dict) else \
        (lambda k: getattr(network_cfg, k))
    return GraphNet(input_dims=list(get("input_dims")), hidden_dim=get("hidden_dim"),
                    output_dim=get("output_dim"), encoder_layers=get("encoder_layers"),
                    decoder_layers=get("decoder_layers"), dropout_rate=get("dropout_rate"),
                    knn_k=get("knn_k"), backbone=get("backbone"), use_mha=get("use_mha"),
                    num_mha_heads=get("num_mha_heads"), mode=get("mode"), conv_module=conv_module)


EVERYDAY_NETWORK = dict(input_dims=[21, 25], use_mha=True, num_mha_heads=2, hidden_dim=256,
                        output_dim=3, encoder_layers=2, decoder_layers=3, dropout_rate=0.0,
                        knn_k=7, backbone="TAGConv", mode="res")  # configs/everyday.json:36-47
