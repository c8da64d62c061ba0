"""How does the CPU baseline (oracle op sequence, B=32 encoder fwd+bwd) scale with threads on this host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deformcontact_amd import synth
from deformcontact_amd.graphnet import ContactEncoder
from oracle import pyg_ref
rest, _, rig = synth.make_batch(32)
torch.manual_seed(0)
enc = ContactEncoder([21, 25], 256, conv_module=pyg_ref)
g_rest = torch.randn(rest.x.shape[0], 256); g_rig = torch.randn(rig.x.shape[0], 256)
def one():
    enc.zero_grad(set_to_none=True)
    a, b = enc(rest, rig)
    torch.autograd.backward([a, b], [g_rest, g_rig])
print("cpu_count", os.cpu_count())
for th in (1, 8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    one()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
    print(th, "threads:", f"{min(ts)*1e3:.0f} ms")
