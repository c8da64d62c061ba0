#!/usr/bin/env python3
"""Randomised shape sweep of the conv layer and the blocked attention against float64 (GPU box).

    python tools/fuzz_parity.py

TAGConv: random N (incl. 1 and non-multiples of the tile), Fi, Fo (incl. 3, 21, 260), K, random
multigraphs - forward, input gradient and every weight gradient vs a dense float64 evaluation.
attention_core: ragged Ns / Nr / d / dv / block sizes - output and the three gradients; then d = dv = 256 (the flash-style
kernels) with query / key counts around the tile edges and score scales from nearly uniform to nearly one-hot weights.  (With a
single key the softmax is constant and dK is exactly zero: its "relative" error is noise / 0.)"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deformcontact_amd as dc
from deformcontact_amd import ops
from deformcontact_amd.graph import GraphIndex
from deformcontact_amd.attention import attention_core
DEV = "cuda:0"
random.seed(1); torch.manual_seed(1)
worst = 0.0
for it in range(40):
    n = random.choice([1, 7, 63, 64, 65, 127, 129, 500, 1000, 2049, 4097])
    fi = random.choice([16, 32, 48, 64, 128, 256, 272])
    fo = random.choice([3, 8, 16, 21, 64, 100, 128, 256, 260])
    k = random.choice([1, 2, 3])
    e = random.randint(0, 6 * n)
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))]).to(DEV)
    conv = dc.nn.TAGConv(fi, fo, K=k).to(DEV)
    x = torch.randn(n, fi, device=DEV, requires_grad=True)
    out = conv(x, ei)
    go = torch.randn_like(out)
    out.backward(go)
    # float64 reference on CPU: dense normalized adjacency
    A = torch.zeros(n, n, dtype=torch.float64)
    src, dst = ei.cpu()
    deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(e, dtype=torch.float64))
    dis = deg.pow(-0.5); dis[torch.isinf(dis)] = 0
    A.index_put_((dst, src), dis[src] * dis[dst], accumulate=True)
    xd = x.detach().double().cpu().requires_grad_()
    ws = [l.weight.detach().double().cpu().requires_grad_() for l in conv.lins]
    xs = [xd]
    for j in range(k): xs.append(A @ xs[-1])
    ref = sum(xs[j] @ ws[j].t() for j in range(k + 1)) + conv.bias.detach().double().cpu()
    ref.backward(go.double().cpu())
    def rel(a, b):
        d = np.abs(b).max(); d = d if d > 0 else 1.0
        return float(np.abs(a - b).max() / d)
    errs = [rel(out.detach().cpu().numpy(), ref.detach().numpy()), rel(x.grad.cpu().numpy(), xd.grad.numpy())]
    errs += [rel(l.weight.grad.cpu().numpy(), w.grad.numpy()) for l, w in zip(conv.lins, ws)]
    worst = max(worst, max(errs))
    if max(errs) > 1e-5:
        print("FAIL", it, n, fi, fo, k, e, errs)
print("tagconv fuzz worst rel err", worst)
worst = 0.0
for it in range(12):
    ns, nr = random.choice([1, 33, 500, 3000]), random.choice([1, 15, 16, 17, 700, 2500])
    d, dv = random.choice([16, 64, 256]), random.choice([16, 64, 256])
    q = (torch.randn(ns, d, device=DEV) * 0.3).requires_grad_(); kk = (torch.randn(nr, d, device=DEV) * 0.3).requires_grad_()
    v = torch.randn(nr, dv, device=DEV).requires_grad_(); go = torch.randn(ns, dv, device=DEV)
    o = attention_core(q, kk, v, block_rows=random.choice([16, 256, 2048])); o.backward(go)
    qd, kd, vd = (t.detach().double().cpu().requires_grad_() for t in (q, kk, v))
    r = torch.softmax(qd @ kd.t(), -1) @ vd; r.backward(go.double().cpu())
    def rel(a, b):
        d_ = float(b.abs().max()); d_ = d_ if d_ > 0 else 1.0
        return float((a.double().cpu() - b).abs().max() / d_)
    errs = [rel(o.detach(), r.detach()), rel(q.grad, qd.grad), rel(kk.grad, kd.grad), rel(v.grad, vd.grad)]
    worst = max(worst, max(errs))
    if max(errs) > 2e-5: print("ATTN FAIL", ns, nr, d, dv, errs)
print("attention fuzz worst rel err", worst)
# the flash-style kernels (d = dv = 256): ragged query / key counts around the 128-query and 32-key tile edges
worst = 0.0
for it in range(16):
    ns = random.choice([1, 31, 127, 128, 129, 255, 700, 2049, 5000]); nr = random.choice([1, 31, 32, 33, 95, 1023, 1024, 1500, 3100])
    sc = random.choice([0.05, 0.3, 1.0])
    q = (torch.randn(ns, 256, device=DEV) * sc).requires_grad_(); kk = (torch.randn(nr, 256, device=DEV) * sc).requires_grad_()
    v = torch.randn(nr, 256, device=DEV).requires_grad_(); go = torch.randn(ns, 256, device=DEV)
    o = attention_core(q, kk, v); o.backward(go)
    qd, kd, vd = (t.detach().double().cpu().requires_grad_() for t in (q, kk, v))
    r = torch.softmax(qd @ kd.t(), -1) @ vd; r.backward(go.double().cpu())
    errs = [rel(o.detach(), r.detach()), rel(q.grad, qd.grad), rel(kk.grad, kd.grad), rel(v.grad, vd.grad)]
    if nr == 1: errs = [errs[0], errs[1], 0.0, errs[3]]      # constant softmax: dK is exactly zero
    worst = max(worst, max(errs))
    if max(errs) > 5e-5: print("FLASH FAIL", ns, nr, sc, errs)
print("flash attention fuzz worst rel err", worst)
