#!/usr/bin/env python3
"""Where does the blocked attention lose precision?  q, k, v as the h=32 golden model produces them;
o / dq / dk / dv of attention_core against float64, next to a plain fp32 evaluation (torch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deformcontact_amd import attention  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model  # noqa: E402
from oracle.weights import fill_state_dict_  # noqa: E402
from tests.helpers import golden_graphs, load_golden  # noqa: E402

dev = "cuda:0"
z = load_golden("graphnet_tag_h32.npz")
m = load_model(dict(EVERYDAY_NETWORK, hidden_dim=int(z["hidden"]), backbone="TAGConv"))
fill_state_dict_(m)
m = m.to(dev)
rest, rig = golden_graphs(z, dev)
with torch.no_grad():
    xs, xr = m.encode(rest, rig)
    head = m.multihead_attention.attention_heads[1]
    q0, k0 = torch.nn.functional.linear(xs, head.weight, head.bias), torch.nn.functional.linear(xr, head.weight, head.bias)
v0 = xr.clone()
go = torch.randn(q0.shape[0], v0.shape[1], device=dev, generator=torch.Generator(device=dev).manual_seed(0))
print("shapes", q0.shape, k0.shape, v0.shape, "|q|max", float(q0.abs().max()), "|k|max", float(k0.abs().max()))


def ref(dtype):
    q, k, v = (t.detach().to(dtype).requires_grad_(True) for t in (q0, k0, v0))
    s = q @ k.t()
    o = torch.softmax(s, -1) @ v
    o.backward(go.to(dtype))
    return [t.detach().double().cpu().numpy() for t in (o, q.grad, k.grad, v.grad)], torch.softmax(s, -1).amax(1)


truth, pmax = ref(torch.float64)
print("softmax row max: median %.3f  >0.99: %d of %d" % (float(pmax.median()), int((pmax > 0.99).sum()), pmax.numel()))
f32, _ = ref(torch.float32)


xs64, xr64 = xs.double().cpu().numpy(), xr.double().cpu().numpy()


def wgrad(got):
    """head.weight gradient the shared Linear receives: dq^T x_soft + dk^T x_rigid (float64 sums)"""
    return got[1].T @ xs64 + got[2].T @ xr64


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def run(label):
    q, k, v = (t.detach().clone().requires_grad_(True) for t in (q0, k0, v0))
    o = attention.attention_core(q, k, v)
    o.backward(go)
    got = [t.detach().double().cpu().numpy() for t in (o, q.grad, k.grad, v.grad)]
    print(f"{label:28s} " + "  ".join(f"{n} {rel(g, t):.2e}" for n, g, t in zip(("o", "dq", "dk", "dv"), got, truth))
          + f"  W.grad {rel(wgrad(got), wgrad(truth)):.2e}  sum_j dk {np.abs(got[2].sum(0)).max():.2e} (truth {np.abs(truth[2].sum(0)).max():.2e})")


print(f"{'torch fp32 (materialised)':28s} " + "  ".join(f"{n} {rel(g, t):.2e}" for n, g, t in zip(("o", "dq", "dk", "dv"), f32, truth))
      + f"  W.grad {rel(wgrad(f32), wgrad(truth)):.2e}  sum_j dk {np.abs(f32[2].sum(0)).max():.2e}")
for es, ea in ((False, False), (True, False), (True, True)):
    attention.EXACT_SCORES, attention.EXACT_ALL = es, ea
    run(f"blocked exact_scores={int(es)} all={int(ea)}")
