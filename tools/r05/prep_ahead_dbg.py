#!/usr/bin/env python3
"""Which part of prepare_weights_ahead makes hipGraph capture of the two-stream encoder step die at capture_end?
Parent (no GPU use) runs each variant in a child process."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

VARIANTS = ["base_off", "on", "on_fwd_only", "on_serial"]


def child(v):
    import torch
    from deformcontact_amd import ops, synth
    from deformcontact_amd.graphnet import ContactEncoder
    dev = torch.device("cuda:0")
    rest, _, rig = (b.to(dev) for b in synth.make_batch(3, soft_vertices=300, sphere_resolution=8))
    g_s = torch.randn(rest.x.shape[0], 256, device=dev)
    g_r = torch.randn(rig.x.shape[0], 256, device=dev)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(dev)
    enc.overlap_branches = v != "on_serial"
    enc.prepare_weights_ahead = v != "base_off"
    if v == "on_no_record_stream":
        torch.Tensor.record_stream_orig = torch.Tensor.record_stream
        take = ops._take_prepared

        def take2(weights, d):
            if not ops._PREPARED:
                return None
            ent = ops._PREPARED.pop(ops._prepared_key(weights), None)
            if ent is not None:
                torch.cuda.current_stream(d).wait_event(ent.event)
            return ent
        ops._take_prepared = take2
    if v == "on_shared_helper":
        ContactEncoder._helper_stream = classmethod(lambda cls, device, which: cls._helper_streams.setdefault(
            "one", torch.cuda.Stream(device=device)))
    if v == "on_wait_stream_only":
        def take3(weights, d):
            if not ops._PREPARED:
                return None
            ent = ops._PREPARED.pop(ops._prepared_key(weights), None)
            if ent is not None:
                cur = torch.cuda.current_stream(d)
                cur.wait_stream(ent.stream)
                for t in ent.tensors():
                    t.record_stream(cur)
            return ent
        ops._take_prepared = take3

    def step():
        a, b = enc(rest, rig)
        if v != "on_fwd_only":
            torch.autograd.backward([a, b], [g_s, g_r])
        return a, b

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(v, "eager ok", flush=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    print(v, "captured", flush=True)
    gr.replay()
    torch.cuda.synchronize()
    print(v, "replayed OK", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for v in VARIANTS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300)
            out = r.stdout.decode().strip().splitlines()
            print(f"{v:24s} rc={r.returncode}  {out[-1] if out else ''}", flush=True)
            if r.returncode:
                err = [l for l in r.stderr.decode().splitlines() if "Extension modules" not in l and "amdgpu.ids" not in l]
                print("   " + "\n   ".join(err[-6:]), flush=True)
