#!/usr/bin/env python3
"""Round 6: which cross-stream dependency patterns does hipGraph capture (ROCm 7.2, torch 2.10) survive?  Each pattern runs in a
child process (a crash in capture_end is a segmentation fault).   python tools/r06/capture_patterns.py [pattern]"""
import subprocess
import sys

import torch


def run(pattern):
    dev = torch.device("cuda:0")
    a = torch.zeros(1 << 20, device=dev)
    b = torch.zeros(1 << 20, device=dev)
    side = torch.cuda.Stream()
    main = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(16)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        a.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            side.wait_stream(main)                      # fork
            a.add_(1)
            with torch.cuda.stream(side):
                b.add_(1)
            if pattern == "fork_join":
                pass
            elif pattern == "main_to_side":             # side waits for an event of main, mid-branch
                evs[0].record(main)
                side.wait_event(evs[0])
                with torch.cuda.stream(side):
                    b.add_(1)
                a.add_(1)
            elif pattern == "side_to_main":
                with torch.cuda.stream(side):
                    evs[0].record(side)
                main.wait_event(evs[0])
                a.add_(1)
                with torch.cuda.stream(side):
                    b.add_(1)
            elif pattern == "both_ways":
                evs[0].record(main)
                side.wait_event(evs[0])
                with torch.cuda.stream(side):
                    b.add_(1)
                    evs[1].record(side)
                main.wait_event(evs[1])
                a.add_(1)
            elif pattern == "ping_pong6":
                for i in range(6):
                    src, dst = (main, side) if i % 2 == 0 else (side, main)
                    with torch.cuda.stream(src):
                        evs[i].record(src)
                    dst.wait_event(evs[i])
                    with torch.cuda.stream(dst):
                        (b if dst is side else a).add_(1)
            elif pattern == "same_event_twice":         # one event object recorded twice in the capture
                evs[0].record(main)
                side.wait_event(evs[0])
                with torch.cuda.stream(side):
                    b.add_(1)
                a.add_(1)
                evs[0].record(main)
                side.wait_event(evs[0])
                with torch.cuda.stream(side):
                    b.add_(1)
            elif pattern == "wait_own_stream":          # a stream waits for its own event
                evs[0].record(main)
                main.wait_event(evs[0])
                a.add_(1)
            main.wait_stream(side)                      # join
            a.add_(1)
        g.replay()
        torch.cuda.synchronize()
    print(pattern, "OK", float(a[0]), float(b[0]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in ("fork_join", "main_to_side", "side_to_main", "both_ways", "ping_pong6", "same_event_twice", "wait_own_stream"):
            r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True)
            print(p, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], (r.stderr.strip().splitlines() or [""])[-1][:120], flush=True)
