#!/usr/bin/env python3
"""Does a hipGraph capture survive nested stream forks (main -> side -> helper -> side -> main) on this ROCm?"""
import os
import subprocess
import sys


def child(v):
    import torch
    dev = torch.device("cuda:0")
    x = torch.randn(1 << 20, device=dev)
    side, helper, helper2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

    def work():
        main = torch.cuda.current_stream()
        if v == "nested":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                helper.wait_stream(side)
                with torch.cuda.stream(helper):
                    a = x * 2
                b = x + 1
                side.wait_stream(helper)
                c = a + b
            main.wait_stream(side)
            return c
        if v == "nested_event":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                helper.wait_stream(side)
                with torch.cuda.stream(helper):
                    a = x * 2
                    ev = torch.cuda.Event()
                    ev.record(helper)
                b = x + 1
                side.wait_event(ev)
                c = a + b
                side.wait_stream(helper)
            main.wait_stream(side)
            return c
        if v == "siblings":
            side.wait_stream(main)
            helper.wait_stream(main)
            with torch.cuda.stream(helper):
                a = x * 2
            with torch.cuda.stream(side):
                b = x + 1
                side.wait_stream(helper)
                c = a + b
            main.wait_stream(side)
            return c
        if v == "two_helpers_nested_and_flat":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                helper.wait_stream(side)
                with torch.cuda.stream(helper):
                    a = x * 2
                b = x + 1
                side.wait_stream(helper)
                c = a + b
            helper2.wait_stream(main)
            with torch.cuda.stream(helper2):
                d = x * 3
            e = x - 1
            main.wait_stream(helper2)
            f = d + e
            main.wait_stream(side)
            return c + f
        raise SystemExit("?")

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ref = work().clone()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = work()
    gr.replay()
    torch.cuda.synchronize()
    print(v, "OK", bool(torch.equal(out, ref)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for v in ["nested", "nested_event", "siblings", "two_helpers_nested_and_flat"]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300)
            out = r.stdout.decode().strip().splitlines()
            print(f"{v:30s} rc={r.returncode}  {out[-1] if out else ''}", flush=True)
