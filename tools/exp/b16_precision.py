#!/usr/bin/env python3
"""Where does the full model's gradient error at batch 16 come from?  HIP path (as configured by the environment:
DC_DENSE_F16X2, DC_DENSE_SPLIT, DC_FUSED_ATTN, DC_ATTN_EXACT_SCORES, ...) against the float64 oracle, next to the fp32
oracle's own distance.  usage: [ENV=..] python tools/exp/b16_precision.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deformcontact_amd import synth  # noqa: E402
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model  # noqa: E402
from deformcontact_amd.train import losses  # noqa: E402
from oracle import pyg_ref  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rest, deff, rig = synth.make_batch(batch)
    torch.manual_seed(0)
    ref = load_model(EVERYDAY_NETWORK, conv_module=pyg_ref)
    with torch.no_grad():
        for name, p_ in ref.named_parameters():
            if name.endswith("bias"):
                p_.uniform_(-0.05, 0.05)
    ref64 = load_model(EVERYDAY_NETWORK, conv_module=pyg_ref)
    ref64.load_state_dict(ref.state_dict())
    ref64 = ref64.double()
    gpu = load_model(EVERYDAY_NETWORK)
    gpu.load_state_dict(ref.state_dict())
    gpu = gpu.to("cuda:0")
    cache = os.path.join("/tmp", f"b16_oracle_{batch}.pt")
    if os.path.exists(cache):
        g32, g64 = torch.load(cache, weights_only=False)
    else:
        losses(ref, rest.clone(), deff.clone(), rig.clone(), 1.0)["loss"].backward()
        c64 = [b.clone() for b in (rest, deff, rig)]
        for b_ in c64:
            b_.x, b_.pos = b_.x.double(), b_.pos.double()
        losses(ref64, *c64, 1.0)["loss"].backward()
        g32 = {n: p.grad.numpy() for n, p in ref.named_parameters()}
        g64 = {n: p.grad.numpy() for n, p in ref64.named_parameters()}
        torch.save((g32, g64), cache)
    losses(gpu, *(b.clone().to("cuda:0") for b in (rest, deff, rig)), 1.0)["loss"].backward()
    torch.cuda.synchronize()
    worst = (0, "")
    rows = []
    for n, p in gpu.named_parameters():
        g = p.grad.cpu().numpy()
        eh, eo = rel(g, g64[n]), rel(g32[n], g64[n])
        rows.append((n, eh, eo))
        if eh / max(eo, 1e-12) > worst[0] and eh > 1e-5:
            worst = (eh / eo, n)
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DC_"))
    print(f"[{tag or 'default'}] worst HIP/oracle ratio above 1e-5: {worst[0]:.2f} ({worst[1]}); "
          f"median e_h {np.median([r[1] for r in rows]):.2e}, median e_o {np.median([r[2] for r in rows]):.2e}")
    for n, eh, eo in rows:
        if "0.bias" in n or "attention" in n or "decoder" in n:
            print(f"    {n:50s} e_h {eh:.2e}  e_o {eo:.2e}")


if __name__ == "__main__":
    main()
