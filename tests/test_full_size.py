"""Full-size (BASELINE.json configs[1] / configs[2]) parity cases that round 2 left open ("what is
weak" 2, 3 and "missing" 4): the B=32 step the bench times with DEFAULT initialisation - the ReLU masks exercised at
full size -, the GCN / GAT backbones at B=32, the loss curve of the shipped configuration (hidden 256, batch 4), and
a pin on WHICH dense kernels a default step launches.  Reference: /root/reference/models/model.py:39-50,69-78,
train.py:46-58,71-73."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deformcontact_amd import _lib, loaders, synth
from deformcontact_amd.graph import clear_cache
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, ContactEncoder, load_model
from oracle import pyg_ref
from tests.helpers import G, assert_parity, record_parity, rel_err, row_rel_err, three_way

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def everyday_b32():
    rest, _, rig = synth.make_batch(32)
    return rest, rig


class _MaskedBranch(torch.nn.Module):
    """One encoder branch of the oracle in float64 with the ReLU replaced by GIVEN 0/1 masks (the masks the HIP
    path applied): the float64 truth of the backward the HIP path actually ran."""

    def __init__(self, convs, masks):
        super().__init__()
        self.convs, self.masks = convs, masks
        self.pre = []

    def forward(self, x, edge_index):
        self.pre = []
        for conv, m in zip(self.convs, self.masks):
            h = conv(x, edge_index)
            self.pre.append(h.detach())
            x = h * m
        return x


@pytest.mark.parametrize("backbone,merged", [("TAGConv", False), ("TAGConv", True), ("GCNConv", False),
                                             ("GATConv", False)])
def test_config1_b32_default_init_step_with_relu_masks_vs_oracle(everyday_b32, backbone, merged):
    """The step the bench times: B=32, DEFAULT initialisation (zero biases: half of the 15 M pre-activations are
    negative, a few lie within fp32 rounding of the kink).  Outputs per row against the fp32 oracle / float64.
    Gradients: the fp32 oracle, the float64 oracle and the HIP path may each decide a handful of near-zero
    pre-activations differently, and ONE flipped mask element moves a bias gradient by 1e-4 whatever the precision of
    the sums - so (1) the HIP masks are checked element by element against the float64 pre-activations: every
    disagreement must sit within 1e-5 of its row's scale of the kink, and (2) the gradients are checked, under the
    usual three-way rule, against the float64 backward evaluated WITH THE HIP PATH'S OWN MASKS - i.e. the ReLU
    backward, the masked dW / dX blocks and the transposed hops are exercised at full size on real masks.
    All three values of `backbone` (models/model.py:39); TAGConv on the per-branch AND the merged path."""
    rest, rig = everyday_b32
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256, backbone=backbone)
    if backbone != "TAGConv":
        with torch.no_grad():                   # PyG's zero-initialised biases would put every GCN / GAT
            for name, p_ in enc.named_parameters():      # pre-activation of a zero-sum row ON the kink
                if name.endswith(".bias"):
                    p_.uniform_(-0.05, 0.05)
    ref = ContactEncoder([21, 25], 256, backbone=backbone, conv_module=pyg_ref)
    ref.load_state_dict(enc.state_dict())
    ref64 = ContactEncoder([21, 25], 256, backbone=backbone, conv_module=pyg_ref)
    ref64.load_state_dict(enc.state_dict())
    ref64 = ref64.double()
    enc = enc.to(DEV)
    enc.merge_branches = merged
    assert enc._mergeable(rest.x.to(DEV), rig.x.to(DEV)) == merged
    clear_cache()
    acts = {}
    hooks = [c.register_forward_hook(lambda m, i, o, k=(b, l): acts.__setitem__(k, o.detach()))
             for b, layers in (("s", enc.conv_layers_resting), ("r", enc.conv_layers_rigid))
             for l, c in enumerate(layers) if l == 0]
    gen = torch.Generator().manual_seed(5)
    ga = torch.randn(rest.x.shape[0], 256, generator=gen)
    gb = torch.randn(rig.x.shape[0], 256, generator=gen)
    a, b = enc(rest.clone().to(DEV), rig.clone().to(DEV))
    torch.autograd.backward([a, b], [ga.to(DEV), gb.to(DEV)])
    for h in hooks:
        h.remove()
    hip_act = {("s", 0): acts[("s", 0)].cpu(), ("r", 0): acts[("r", 0)].cpu(), ("s", 1): a.detach().cpu(),
               ("r", 1): b.detach().cpu()}
    assert 0.1 < float((hip_act[("s", 1)] > 0).double().mean()) < 0.9          # the masks are real masks
    # fp32 oracle, its own ReLU
    rest_c, rig_c = G(rest.x, rest.edge_index), G(rig.x, rig.edge_index)
    ra, rb = ref(rest_c, rig_c)
    torch.autograd.backward([ra, rb], [ga, gb])
    # float64 oracle with the HIP path's masks
    br = {"s": _MaskedBranch(ref64.conv_layers_resting, [(hip_act[("s", l)] > 0).double() for l in range(2)]),
          "r": _MaskedBranch(ref64.conv_layers_rigid, [(hip_act[("r", l)] > 0).double() for l in range(2)])}
    ta = br["s"](rest.x.double(), rest.edge_index)
    tb = br["r"](rig.x.double(), rig.edge_index)
    torch.autograd.backward([ta, tb], [ga.double(), gb.double()])
    # (1) mask consistency: where the HIP mask and the float64 pre-activation's sign disagree, the pre-activation
    # is within rounding of zero on its row's scale
    flips = 0
    for k in ("s", "r"):
        for l in range(2):
            pre = br[k].pre[l]
            bad = (hip_act[(k, l)] > 0) != (pre > 0)
            flips += int(bad.sum())
            if bad.any():
                scale = pre.abs().amax(dim=1, keepdim=True).expand_as(pre)
                assert float((pre.abs() / scale)[bad].max()) < 1e-5, f"mask of branch {k} layer {l}"
    assert flips < 2000, f"{flips} mask elements differ from the float64 evaluation"
    # (2) outputs per row, every parameter gradient
    assert_parity(_np(a), _np(ra), _np(ta), TOL, "soft per row", metric=row_rel_err)
    assert_parity(_np(b), _np(rb), _np(tb), TOL, "rigid per row", metric=row_rel_err)
    truth = {n_: p_.grad.clone() for n_, p_ in ref64.named_parameters()}
    # the fp32 oracle's own distance from float64 (both with their own ReLU): what fp32 rounding alone costs
    ref64.zero_grad(set_to_none=True)
    ua, ub = ref64(G(rest.x.double(), rest.edge_index), G(rig.x.double(), rig.edge_index))
    torch.autograd.backward([ua, ub], [ga.double(), gb.double()])
    rp, up = dict(ref.named_parameters()), dict(ref64.named_parameters())
    for name, p in enc.named_parameters():
        e_h = rel_err(_np(p.grad), _np(truth[name]))
        e_o = rel_err(_np(rp[name].grad), _np(up[name].grad))
        # GAT's attention vectors: their gradient is a sum over all nodes of terms that cancel to ~1 % of their
        # size (the softmax weights of a segment sum to one), so it sits 1e-5 from float64 in the fp32 oracle
        # already; the HIP path's dense blocks carry 22 rather than 24 bits - 3 x the oracle's distance there
        special = ("a gradient that is mathematically zero: rounding noise of the softmax backward; bar 3 x the fp32 "
                   "oracle's distance from float64" if name.endswith("att_dst") else
                   "GAT attention vector: a sum of terms that cancel to ~1 % of their size; bar 3 x the fp32 "
                   "oracle's distance from float64" if ".att_" in name else None)
        record_parity(name + " (vs float64 over the HIP masks)", rel_err(_np(p.grad), _np(rp[name].grad)), True, e_h, e_o,
                      special=special)
        bound = TOL if special is None else max(3 * e_o, TOL)
        assert e_h <= bound, (f"{name}: {e_h:.2e} from the float64 backward over the HIP path's own "
                              f"masks (bound {bound:.2e}; fp32 oracle vs float64: {e_o:.2e})")
        # and the north-star bar against the fp32 oracle wherever the two fp32 evaluations took the same masks
        if flips == 0:
            assert_parity(_np(p.grad), _np(rp[name].grad), _np(truth[name]), TOL, name)


def test_config2_loss_curve_of_the_shipped_configuration_vs_oracle():
    """BASELINE.json configs[2] at the SHIPPED configuration (configs/everyday.json: hidden 256, 2 + 2 TAGConv
    layers, 2 attention heads, batch 4; full-size meshes): 8 training steps, a new batch every step, same init,
    Adam(4e-4) - HIP path (FlatAdam over the direct-gradient bucket, steps replayed from one hipGraph) vs the CPU
    oracle in fp32 and in float64.  Per step the three-way rule: within 1e-5 of the fp32 oracle's loss, or no
    further from the float64 curve than twice the fp32 oracle is."""
    from deformcontact_amd import dp
    from deformcontact_amd.train import GraphedTrainStep, train_step
    torch.manual_seed(0)
    ref = load_model(EVERYDAY_NETWORK, conv_module=pyg_ref)
    ref64 = load_model(EVERYDAY_NETWORK, conv_module=pyg_ref)
    ref64.load_state_dict(ref.state_dict())
    ref64 = ref64.double()
    gpu = load_model(EVERYDAY_NETWORK)
    gpu.load_state_dict(ref.state_dict())
    gpu = gpu.to(DEV)
    o_ref = torch.optim.Adam(ref.parameters(), lr=4e-4)
    o_ref64 = torch.optim.Adam(ref64.parameters(), lr=4e-4)
    bucket = dp.GradBucket(gpu.parameters(), direct=True)
    o_gpu = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    step = GraphedTrainStep(gpu, o_gpu, bucket, 1.0, eager_steps=2)
    ds = loaders.SyntheticEverydayDataset(32, 0)
    curves = {"gpu": [], "f32": [], "f64": []}
    for collated in loaders.iterate_batches(ds, 4):
        cpu = loaders.to_batches(collated)
        curves["f32"].append(float(train_step(ref, o_ref, *cpu)["loss"]))
        c64 = loaders.to_batches(collated)
        for bt in c64:
            bt.x, bt.pos = bt.x.double(), bt.pos.double()
        curves["f64"].append(float(train_step(ref64, o_ref64, *c64)["loss"]))
        curves["gpu"].append(float(step(*loaders.to_batches(collated, DEV))["loss"]))
    assert len(curves["gpu"]) == 8 and step.replays == 6
    for i, (g, r, t) in enumerate(zip(curves["gpu"], curves["f32"], curves["f64"])):
        d = abs(g - r) / abs(r)
        record_parity(f"loss of step {i}", d, d >= TOL, abs(g - t) / abs(t), abs(r - t) / abs(t),
                      special=None if i == 0 else f"loss after {i} Adam steps (every parameter has moved ~lr per step)")
        if d >= TOL:
            # step 0 (identical parameters): the plain bar.  Later steps are `special`: every parameter has moved ~lr
            # per step along rounding-dependent directions; bound = twice the fp32 oracle's own drift from float64
            e_h, e_o = abs(g - t) / abs(t), abs(r - t) / abs(t)
            assert e_h <= (TOL if i == 0 else max(2 * e_o, TOL)), (
                f"step {i}: HIP {g:.9g} vs fp32 oracle {r:.9g} ({d:.2e}); vs float64 {t:.9g}: HIP {e_h:.2e}, oracle {e_o:.2e}",
                curves)
    assert curves["gpu"][-1] < curves["gpu"][0]
    # the parameters after 8 steps: against the float64 run, no worse than twice the fp32 oracle's drift
    for (name, pg), pr, pt in zip(gpu.named_parameters(), ref.parameters(), ref64.parameters()):
        e_h = rel_err(_np(pg), _np(pt))
        e_o = rel_err(_np(pr), _np(pt))
        # (Adam moves every element by ~lr per step whatever its gradient's size: an element whose gradient is
        # within rounding of zero walks differently in every evaluation - the fp32 oracle itself ends 5e-4 from
        # the float64 run; the bound says the HIP path's walk is of the same kind, not that it is the same walk)
        record_parity(name + " after 8 Adam steps", rel_err(_np(pg), _np(pr)), True, e_h, e_o, tol=1e-4,
                      special="parameters after 8 Adam steps: every element moves ~lr per step whatever its gradient's size")
        assert e_h <= max(4 * e_o, 1e-4), f"{name}: HIP {e_h:.2e} vs oracle {e_o:.2e} from float64 after 8 steps"


class _ReluShim:
    """Stands in for ``graphnet.F`` while an ORACLE model runs: every ReLU of the reference wiring (the encoder loops
    ``models/model.py:69-78`` and the decoder ``:52-64``) is keyed by (rows of its input, occurrence) and either
    records its own 0/1 mask and pre-activation or applies a GIVEN mask - the masks another evaluation took."""

    def __init__(self, masks=None):
        self.masks, self.own, self.pre, self.count = masks, {}, {}, {}

    def __getattr__(self, name):
        return getattr(F, name)

    def relu(self, x, inplace=False):
        n = x.shape[0]
        key = (n, self.count.get(n, 0))
        self.count[n] = key[1] + 1
        self.pre[key] = x.detach()
        if self.masks is None:
            y = torch.relu(x)
            self.own[key] = y.detach() > 0
            return y
        return x * self.masks[key].to(x.dtype)


_B16 = {}


def _b16_oracles(monkeypatch):
    """Batch 16 at the shipped widths: the models, the batches, the fp32 oracle's step (its own ReLU masks recorded)
    and a function that runs the float64 oracle's step over GIVEN masks (cached: the three variants share them)."""
    from deformcontact_amd import graphnet
    from deformcontact_amd.train import losses
    if not _B16:
        rest, deff, rig = synth.make_batch(16)
        torch.manual_seed(0)
        ref = load_model(EVERYDAY_NETWORK, conv_module=pyg_ref)
        with torch.no_grad():                                               # biases off zero: fewer ReLUs on their kink
            for name, p_ in ref.named_parameters():
                if name.endswith("bias"):
                    p_.uniform_(-0.05, 0.05)
        state = {k: v.clone() for k, v in ref.state_dict().items()}
        shim = _ReluShim()
        monkeypatch.setattr(graphnet, "F", shim)
        pos = {}
        h = ref.register_forward_hook(lambda mod, i, o: pos.__setitem__("pos", o.pos.detach().double().numpy()))
        o32 = losses(ref, rest.clone(), deff.clone(), rig.clone(), 1.0)
        o32["loss"].backward()
        h.remove()
        monkeypatch.setattr(graphnet, "F", F)
        _B16.update(batches=(rest, deff, rig), state=state, masks32=shim.own, pos32=pos["pos"],
                    loss32={k: float(v.detach()) for k, v in o32.items()},
                    grad32={n: _np(p.grad) for n, p in ref.named_parameters()}, truth={})

    def float64_over(masks, tag):
        key = (tag, tuple(sorted((k, int(m.sum())) for k, m in masks.items())))
        if key not in _B16["truth"]:
            m64 = load_model(EVERYDAY_NETWORK, conv_module=pyg_ref)
            m64.load_state_dict(_B16["state"])
            m64 = m64.double()
            shim = _ReluShim(masks)
            monkeypatch.setattr(graphnet, "F", shim)
            pos = {}
            h = m64.register_forward_hook(lambda mod, i, o: pos.__setitem__("pos", o.pos.detach().numpy()))
            c64 = [b.clone() for b in _B16["batches"]]
            for b_ in c64:
                b_.x, b_.pos = b_.x.double(), b_.pos.double()
            o64 = losses(m64, *c64, 1.0)
            o64["loss"].backward()
            h.remove()
            monkeypatch.setattr(graphnet, "F", F)
            _B16["truth"][key] = dict(pos=pos["pos"], loss={k: float(v.detach()) for k, v in o64.items()}, pre=shim.pre,
                                      grad={n: _np(p.grad) for n, p in m64.named_parameters()})
        return _B16["truth"][key]
    return _B16, float64_over


@pytest.mark.parametrize("variant", ["one_sweep", "two_sweeps", "blocked_backward"])
def test_full_model_b16_through_the_flash_attention_vs_oracle(variant, monkeypatch):
    """VERDICT r03 "what is weak" 2: the path `full_train_step_b32` times had never been compared with the oracle AS A
    MODEL.  GraphNet at the shipped widths (hidden 256, 2 heads; configs/everyday.json:36-47), batch 16 - 16,384 x
    12,192 scores >= 2^26, so `CrossAttention` takes `attention_core` -> `dc_attn_flash_fwd` and, by default, the
    one-sweep `dc_attn_flash_ds` backward with the eps-corrected dQ / dK - one training step's forward + both losses
    + backward (/root/reference/models/model.py:82-95, train.py:46-58), also with the two-sweep form and with the
    blocked backward (P + dS budget forced below what batch 16 needs).
    Prediction and losses: against the fp32 oracle (float64 where they differ by more than 1e-5).  Gradients: of the
    9 M ReLU pre-activations of a step a few lie within rounding of zero, and ONE element decided differently moves the
    gradients behind it by 5e-5 whatever the precision of the sums (seen here: decoder layer 6 between the fp32 oracle and
    float64, decoder layer 3 between the flash forward and float64).  So, as in the B = 32 encoder test above: (1) the HIP
    path's masks are checked element by element against the float64 pre-activations, (2) EVERY parameter gradient is
    held to the float64 backward evaluated over the HIP path's own masks, no further from it than twice the fp32 oracle
    is from the float64 backward over ITS masks (or 1e-5), and (3) where the HIP path and the fp32 oracle took the same
    masks, to 1e-5 of the fp32 oracle under the usual three-way rule."""
    from deformcontact_amd import attention, graphnet
    from deformcontact_amd.train import losses
    ctx, float64_over = _b16_oracles(monkeypatch)
    rest, deff, rig = ctx["batches"]
    ns, nr = rest.x.shape[0], rig.x.shape[0]
    assert ns * nr >= 1 << 26                                              # the library attention is what runs
    monkeypatch.setattr(attention, "FLASH", True)
    monkeypatch.setattr(attention, "FLASH_BWD", True)
    monkeypatch.setattr(attention, "FLASH_BWD_SINGLE", variant != "two_sweeps")
    if variant == "blocked_backward":
        monkeypatch.setattr(attention, "FLASH_BWD_MAX_BYTES", 1 << 20)
    gpu = load_model(EVERYDAY_NETWORK)
    gpu.load_state_dict(ctx["state"])
    gpu = gpu.to(DEV)
    clear_cache()
    calls = {"flash_fwd": 0, "flash_ds": 0}
    L = _lib.lib()

    class Spy:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)
    monkeypatch.setattr(L, "dc_attn_flash_fwd", Spy(L.dc_attn_flash_fwd, "flash_fwd"), raising=False)
    monkeypatch.setattr(L, "dc_attn_flash_ds", Spy(L.dc_attn_flash_ds, "flash_ds"), raising=False)
    # the HIP path's ReLU masks: encoder layers (fused into the dense epilogue) and decoder layers
    masks, dec = {}, []
    hooks = [c.register_forward_hook(lambda m, i, o, k=k: masks.__setitem__(k, (o.detach() > 0).cpu()))
             for k, c in (((ns, 0), gpu.conv_layers_resting[0]), ((ns, 1), gpu.conv_layers_resting[1]),
                          ((nr, 0), gpu.conv_layers_rigid[0]), ((nr, 1), gpu.conv_layers_rigid[1]))]
    real_linear = graphnet._linear

    def spy_linear(lin, x, relu=False):
        y = real_linear(lin, x, relu)
        if relu:
            dec.append((y.detach() > 0).cpu())
        return y
    monkeypatch.setattr(graphnet, "_linear", spy_linear)
    pos = {}
    hooks.append(gpu.register_forward_hook(lambda mod, i, o: pos.__setitem__("pos", o.pos.detach().cpu().double().numpy())))
    out = losses(gpu, *(b.clone().to(DEV) for b in (rest, deff, rig)), 1.0)
    out["loss"].backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    monkeypatch.setattr(graphnet, "_linear", real_linear)
    assert calls["flash_fwd"] == 2 and calls["flash_ds"] == (0 if variant == "blocked_backward" else 2), calls
    assert len(dec) == 3
    masks.update({(ns, 2 + i): m for i, m in enumerate(dec)})
    assert set(masks) == set(ctx["masks32"])
    t_h, t_o = float64_over(masks, "hip"), float64_over(ctx["masks32"], "f32")
    # (1) mask consistency with float64: every disagreement within 1e-5 of its row's scale of the kink
    flips = same_as_f32 = 0
    for k, m in masks.items():
        pre = t_h["pre"][k]
        bad = m != (pre > 0)
        flips += int(bad.sum())
        same_as_f32 += int((m != ctx["masks32"][k]).sum())
        if bad.any():
            scale = pre.abs().amax(dim=1, keepdim=True).expand_as(pre)
            assert float((pre.abs() / scale)[bad].max()) < 1e-5, f"ReLU {k}: a mask element far from the kink differs"
    assert flips < 200, f"{flips} mask elements differ from the float64 evaluation"
    # prediction and losses
    three_way(rel_err(pos["pos"], ctx["pos32"]), lambda: rel_err(pos["pos"], t_h["pos"]),
              lambda: rel_err(ctx["pos32"], t_o["pos"]), TOL, f"{variant}: pred.pos")
    for key in ("loss", "l1", "consistency"):
        g, r = float(out[key].detach()), ctx["loss32"][key]
        three_way(abs(g - r) / abs(r), lambda: abs(g - t_h["loss"][key]) / abs(t_h["loss"][key]),
                  lambda: abs(r - t_o["loss"][key]) / abs(t_o["loss"][key]), TOL, f"{variant}: {key}")
    # (2) + (3) every parameter gradient
    for name, p in gpu.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        got = _np(p.grad)
        e_h, e_o = rel_err(got, t_h["grad"][name]), rel_err(ctx["grad32"][name], t_o["grad"][name])
        d = rel_err(got, ctx["grad32"][name])
        record_parity(f"{variant}: grad {name} (float64 over each path's own masks; {same_as_f32} mask elements differ "
                      f"between HIP and the fp32 oracle)", d, d >= TOL, e_h, e_o)
        assert e_h <= TOL, (f"{name}: {e_h:.2e} from the float64 backward over the HIP path's own masks "
                            f"(fp32 oracle vs float64 over its masks: {e_o:.2e})")


@pytest.mark.parametrize("merged", [False, True])
def test_default_b32_step_launches_no_generic_dense_kernel(everyday_b32, merged):
    """Pin on the kernel surface: a default-config B=32 encoder step (forward + backward, direct-gradient bucket)
    and a shipped-config full step at batch 4 never reach the generic (bounds-checked, scalar-load) dense kernels
    (`dc_generic_dense_launches`; round 2 found two silent fallbacks of that kind by accident)."""
    from deformcontact_amd import dp
    from deformcontact_amd.train import losses
    L = _lib.lib()
    rest, rig = everyday_b32
    rest, rig = rest.clone().to(DEV), rig.clone().to(DEV)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(DEV)
    enc.merge_branches = merged
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    bucket.zero()
    ga = torch.randn(rest.x.shape[0], 256, device=DEV)
    gb = torch.randn(rig.x.shape[0], 256, device=DEV)
    clear_cache()
    L.dc_generic_dense_launches(1)
    a, b = enc(rest, rig)
    torch.autograd.backward([a, b], [ga, gb])
    torch.cuda.synchronize()
    assert L.dc_generic_dense_launches(1) == 0, "a dense block of the default B=32 step took the generic kernel"
    # the whole reference step at the shipped batch size (decoder and attention heads included)
    model = load_model(EVERYDAY_NETWORK).to(DEV)
    model.merge_branches = merged
    mb = dp.GradBucket(model.parameters(), direct=True)
    mb.zero()
    r4, d4, g4 = (t.to(DEV) for t in synth.make_batch(4))
    losses(model, r4, d4, g4, 1.0)["loss"].backward()
    torch.cuda.synchronize()
    n = L.dc_generic_dense_launches(1)
    # the decoder's last Linear (256 -> 3) has no tuned kernel: Fo = 3.  Its forward, dX and dW are the only
    # generic launches of the step.
    assert n <= 3, f"{n} generic dense launches in the shipped-config step (expected the 256 -> 3 output layer only)"


def test_default_b32_step_launches_exactly_the_tuned_kernels(everyday_b32):
    """WHICH kernels a default-config B=32 encoder step (new batch: adjacency build + forward +
    backward, direct-gradient bucket) launches, by name (`dc_kernel_trace`): the one-launch segmented adjacency build, the
    fused pack + narrow hops of the first layers, the 3-hop chain launches (`k_hop_chain_gcn`, not 12 `k_spmm_wave`
    launches), the 128 x 256-tile fp16x2 blocks (`k_fwd_h2d`, `k_dw_h2w`), the six-product narrow blocks - and none
    of the generic / fallback kernels."""
    from deformcontact_amd import dp
    rest, rig = everyday_b32
    rest, rig = rest.clone().to(DEV), rig.clone().to(DEV)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(DEV)
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    bucket.zero()
    ga = torch.randn(rest.x.shape[0], 256, device=DEV)
    gb = torch.randn(rig.x.shape[0], 256, device=DEV)
    clear_cache()
    torch.cuda.synchronize()
    _lib.kernel_trace(True)
    a, b = enc(rest, rig)
    torch.autograd.backward([a, b], [ga, gb])
    torch.cuda.synchronize()
    _lib.kernel_trace(False)
    got = _lib.kernel_trace_counts()
    names = {k.split("<")[0] for k in got}
    want = {
        "k_build_segment": 2,                # both sorted adjacencies + gcn_norm of a graph: ONE launch per graph
        "k_hop_chain_gcn<8>": 2,             # soft: forward chain + transposed chain of layer 2
        "k_hop_chain_gcn<6>": 2,             # rigid
        "k_weight_prep": 2,                  # layer-2 weights (+ clears the chain's row maxima)
        "k_fwd_h2d<true>": 2,                # soft (whole 128-row tiles): layer-2 forward block (bias + ReLU epilogue)
        "k_fwd_h2d<false>": 2,               # and dX as a forward-shaped block over the gradient slab; rigid (ragged)
        "k_dw_h2w<false>": 2,                # layer-2 dW
        "k_mask_grad": 2,
    }
    for k, v in want.items():
        assert got.get(k) == v, (k, got)
    # first layers: fused pack + first hop, two more narrow hops, six-product split blocks
    narrow_hops = sum(v for k, v in got.items() if k.startswith("k_spmm_sub"))
    assert narrow_hops == 6, got
    assert sum(v for k, v in got.items() if k.startswith("k_fwd_narrow")) == 2       # (round 6: the short-reduction kernel;
    assert not any(k.startswith("k_fwd_split") or "k_pack_weights" in k for k in got)  # no packing launch in front of it)
    assert sum(v for k, v in got.items() if k.startswith("k_dw_split")) == 2
    assert "k_dw_reduce" in names
    banned = {"k_spmm_wave", "k_tag_linear_fwd", "k_tag_linear_bwd_dx", "k_tag_linear_bwd_dw", "k_fwd_h2", "k_fwd_fast",
              "k_init", "k_count", "k_fill", "k_emit", "k_hop_chain", "k_fwd_h2w"}
    assert not (names & banned), (names & banned, got)


@pytest.mark.parametrize("npad,n", [(8192 + 4, 8192 + 1), (16384 + 128, 16384 + 77), (24448, 24384), (32768, 32000)])
def test_attention_row_kernels_register_rows_vs_strided_and_float64(npad, n):
    """the batch-32 step launches the register-row variants of the attention row kernels
    (`k_attn_softmax_rows_reg<16/24/32>`, `k_attn_ds_rows_reg<16/24>`: rows of up to 32 K floats held in registers)
    which no unit test reached.  Each against the strided kernels (forced through a 4-byte-misaligned copy of the
    same rows) and against the float64 formulas, with padded key columns (n < npad)."""
    from deformcontact_amd.graph import current_stream_ptr
    L = _lib.lib()
    st = current_stream_ptr(torch.device(DEV))
    rows = 6
    gen = torch.Generator().manual_seed(npad)
    s0 = (torch.randn(rows, npad, generator=gen) * 3.0).to(DEV)
    dp0 = torch.randn(rows, npad, generator=gen).to(DEV)

    def misaligned(t):                                     # same values, row pointer 4 bytes off a 16-byte boundary
        buf = torch.empty(rows * npad + 8, device=DEV)
        v = buf[1:1 + rows * npad].view(rows, npad)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v
    res = {}
    for tag, mk in (("reg", lambda t: t.clone()), ("strided", misaligned)):
        s, lse = mk(s0), torch.empty(rows, device=DEV)
        _lib.check(L.dc_attn_softmax_rows(s.data_ptr(), npad, rows, n, npad, lse.data_ptr(), st), "softmax")
        dpv, rm = mk(dp0), torch.empty(rows, device=DEV)
        _lib.check(L.dc_attn_ds_rows(s.data_ptr(), dpv.data_ptr(), npad, rows, npad, None, rm.data_ptr(), st), "ds")
        torch.cuda.synchronize()
        res[tag] = (s.clone(), lse, dpv.clone(), rm)
    s64 = s0[:, :n].double().cpu()
    p64 = torch.softmax(s64, dim=1)
    lse64 = torch.logsumexp(s64, dim=1)
    d64 = dp0[:, :n].double().cpu()
    delta = (p64 * d64).sum(1, keepdim=True) / p64.sum(1, keepdim=True)
    ds64 = p64 * (d64 - delta)
    for tag, (p, lse, ds, rm) in res.items():
        assert not p[:, n:].any() and not ds[:, n:].any(), f"{tag}: padded key columns must come out zero"
        assert rel_err(_np(p[:, :n]), p64.numpy()) < 2e-6, tag
        assert np.abs(_np(lse) - lse64.numpy()).max() < 2e-6 * np.abs(lse64.numpy()).max(), tag
        assert rel_err(_np(ds[:, :n]), ds64.numpy()) < 5e-6, tag
        assert np.allclose(_np(rm), np.abs(_np(ds)).max(1), rtol=0, atol=0), tag
    assert rel_err(_np(res["reg"][0]), _np(res["strided"][0])) < 1e-6
    assert rel_err(_np(res["reg"][2]), _np(res["strided"][2])) < 2e-6
