#!/usr/bin/env python3
"""Drive tools/exp/libhopexp.so (experimental hop variants) on the B=32 everyday graphs.

    python tools/exp/hop_exp.py [--reps 200]

Prints us per launch and the compulsory-bytes fraction of 8 TB/s for every variant, on the slab
column-block layout a step uses (ld = 1024) and on contiguous [N,256] buffers, checks each
variant bit-for-bit against the product kernel, and times the sorted-adjacency build."""
import argparse
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

from deformcontact_amd import ops, synth  # noqa: E402
from deformcontact_amd.graph import GraphIndex, current_stream_ptr  # noqa: E402

NAMES = {0: "copy rows (floor)", 1: "chunk RW1 WPB4 (= product)", 2: "chunk RW1 nt-store", 3: "chunk RW2",
         4: "chunk RW4", 5: "chunk RW8", 6: "chunk RW4 WPB8", 7: "chunk RW1 WPB16", 8: "pipe RW4",
         9: "pipe RW8", 10: "pipe RW16", 11: "pipe RW8 nt", 12: "pipe RW4 WPB8", 13: "pipe RW8 WPB2",
         14: "chunk RW1 U4", 15: "pipe RW4 U6", 16: "swp RW2", 17: "swp RW4", 18: "swp RW8", 19: "swp RW16"}


def timeit(fn, reps):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    so = os.path.join(HERE, "libhopexp.so")
    if not os.path.exists(so):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
                               "-ffp-contract=off", os.path.join(HERE, "hop_exp.hip"), "-o", so])
    L = ctypes.CDLL(so)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    L.hopexp_run.argtypes = [ctypes.c_int, vp, vp, vp, vp, i64, vp, i64, i64, vp]
    dev = torch.device("cuda:0")
    st = current_stream_ptr(dev)
    rest, _, rig = synth.make_batch(32)
    f = 256
    for name, b in (("soft", rest), ("rigid", rig)):
        n, e = b.x.shape[0], b.edge_index.shape[1]
        g = GraphIndex(b.edge_index.to(dev), n)
        comp = e * 8 + n * (8 * f + 4)
        slab = torch.randn(n, 4 * f, device=dev)
        xc, yc = slab[:, :f].contiguous(), torch.empty(n, f, device=dev)
        for lay, x, y in (("slab ld=1024", slab[:, :f], slab[:, f:2 * f]), ("contiguous", xc, yc)):
            ref = ops.hop(g.fwd, x).clone()
            rm = torch.zeros(n, device=dev)
            t = timeit(lambda: ops.hop(g.fwd, x, out=y), args.reps)
            print(f"{name:5s} {lay:13s} product dc_spmm_f32        : {t * 1e3:7.2f} us  frac {comp / t / 1e6 / 8000:.3f}")
            t = timeit(lambda: ops.hop(g.fwd, x, out=y, rowmax=rm, rowmax_mode=2), args.reps)
            print(f"{name:5s} {lay:13s} product dc_spmm_f32_rowmax : {t * 1e3:7.2f} us  frac {comp / t / 1e6 / 8000:.3f}")
            for v in sorted(NAMES):
                def run(v=v):
                    rc = L.hopexp_run(v, g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr(),
                                      x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), n, st)
                    assert rc == 0, rc
                y.zero_()
                run()
                torch.cuda.synchronize()
                ok = "exact" if (v == 0 or torch.equal(y, ref)) else "MISMATCH"
                t = timeit(run, args.reps)
                print(f"{name:5s} {lay:13s} v{v:02d} {NAMES[v]:26s}: {t * 1e3:7.2f} us  frac {comp / t / 1e6 / 8000:.3f}  {ok}")
        # packed-record variants: rec = CSR-ordered {idx, w} (+8 padding); rec8 = 8 records per row
        ptr_c, oth_c, w_c = g.fwd.ptr.cpu().long(), g.fwd.other.cpu(), g.fwd.w.cpu()
        rec = torch.zeros(e + 8, 2, dtype=torch.int32)
        rec[:e, 0] = oth_c[:e]
        rec[:e, 1] = w_c[:e].view(torch.int32)
        pos = ptr_c[:-1, None] + torch.arange(8)[None, :]
        ok = pos < ptr_c[1:, None]
        rec8 = torch.zeros(n, 8, 2, dtype=torch.int32)
        rec8[ok] = rec[pos[ok]]
        rec_d, rec8_d = rec.to(dev), rec8.to(dev)
        L.hopexp_run_rec.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, i64, i64, vp]
        for lay, x, y in (("slab ld=1024", slab[:, :f], slab[:, f:2 * f]),):
            ref = ops.hop(g.fwd, x).clone()
            for v, nm, rr in ((20, "rec csr", rec_d), (21, "rec csr clamp", rec_d), (22, "rec ELL8", rec8_d),
                              (23, "rec ELL8 clamp", rec8_d), (24, "rec ELL8 plain store", rec8_d)):
                def run(v=v, rr=rr):
                    rc = L.hopexp_run_rec(v, g.fwd.ptr.data_ptr(), rr.data_ptr(), g.fwd.other.data_ptr(),
                                          g.fwd.w.data_ptr(), x.data_ptr(), x.stride(0), y.data_ptr(),
                                          y.stride(0), n, st)
                    assert rc == 0, rc
                y.zero_()
                run()
                torch.cuda.synchronize()
                okk = "exact" if torch.equal(y, ref) else "MISMATCH"
                t = timeit(run, args.reps)
                print(f"{name:5s} {lay:13s} v{v:02d} {nm:26s}: {t * 1e3:7.2f} us  frac {comp / t / 1e6 / 8000:.3f}  {okk}")
        # chains of 3 hops inside one slab (block j -> j+1), as a TAGConv layer runs them
        for v in (1, 2, 14):
            def chain(v=v):
                for j in range(3):
                    xs, ys = slab[:, j * f:(j + 1) * f], slab[:, (j + 1) * f:(j + 2) * f]
                    L.hopexp_run(v, g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr(),
                                 xs.data_ptr(), xs.stride(0), ys.data_ptr(), ys.stride(0), n, st)
            t = timeit(chain, args.reps)
            print(f"{name:5s} chain of 3 hops in the slab, v{v:02d} {NAMES[v]:26s}: {t * 1e3:7.2f} us  ({t * 1e3 / 3:.2f} per hop)")
        # the step's regime: more slabs alive than the 256 MiB Infinity Cache holds (here 4 x this
        # graph's slab), chains of 3 hops on each in turn - every chain starts from cold rows
        slabs = [torch.randn(n, 4 * f, device=dev) for _ in range(4 if name == "soft" else 5)]
        rms = [torch.zeros(n, device=dev) for _ in slabs]

        def rot(kind):
            for sl, rmx in zip(slabs, rms):
                for j in range(3):
                    xs, ys = sl[:, j * f:(j + 1) * f], sl[:, (j + 1) * f:(j + 2) * f]
                    if kind == "plain":
                        ops.hop(g.fwd, xs, out=ys)
                    elif kind == "rowmax":
                        ops.hop(g.fwd, xs, out=ys, rowmax=rmx, rowmax_mode=1 if j == 0 else 2)
                    elif kind >= 20:
                        rr = rec8_d if kind >= 22 else rec_d
                        L.hopexp_run_rec(kind, g.fwd.ptr.data_ptr(), rr.data_ptr(), g.fwd.other.data_ptr(),
                                         g.fwd.w.data_ptr(), xs.data_ptr(), xs.stride(0), ys.data_ptr(),
                                         ys.stride(0), n, current_stream_ptr(dev))
                    else:
                        L.hopexp_run(kind, g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr(),
                                     xs.data_ptr(), xs.stride(0), ys.data_ptr(), ys.stride(0), n,
                                     current_stream_ptr(dev))
        # the same chains over SEPARATE contiguous [n, 256] blocks (row stride 1 KiB instead of the slab's 4 KiB + pad)
        cslabs = [[torch.randn(n, f, device=dev) for _ in range(4)] for _ in slabs]

        def rot_contig(kind):
            for blocks in cslabs:
                for j in range(3):
                    xs, ys = blocks[j], blocks[j + 1]
                    if kind == "plain":
                        ops.hop(g.fwd, xs, out=ys)
                    else:
                        L.hopexp_run(kind, g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr(),
                                     xs.data_ptr(), xs.stride(0), ys.data_ptr(), ys.stride(0), n,
                                     current_stream_ptr(dev))
        for kind in ("plain", 0, 1):
            rot_contig(kind)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                rot_contig(kind)
            t = timeit(gr.replay, 20) / (3 * len(cslabs))
            label = kind if isinstance(kind, str) else f"v{kind:02d} {NAMES.get(kind, 'rec variant')}"
            print(f"{name:5s} rotating CONTIGUOUS blocks (> Infinity Cache), graph-replayed chains, {label:32s}: {t * 1e3:7.2f} us per hop  frac {comp / t / 1e6 / 8000:.3f}")
        # windowed variants (R destination rows, halo H) in the same regime, checked bit for bit first
        L.hopexp_run_win.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, i64, vp, i64, i64, vp]
        for R_, H_ in ((64, 40), (60, 48), (64, 56), (96, 56), (128, 56), (128, 40), (32, 56)):
            def rot_win():
                for sl in slabs:
                    for j in range(3):
                        xs, ys = sl[:, j * f:(j + 1) * f], sl[:, (j + 1) * f:(j + 2) * f]
                        rc = L.hopexp_run_win(R_, H_, g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr(),
                                              xs.data_ptr(), xs.stride(0), ys.data_ptr(), ys.stride(0), n,
                                              current_stream_ptr(dev))
                        assert rc == 0, rc
            xs, ys = slabs[0][:, :f], slabs[0][:, f:2 * f]
            ref = ops.hop(g.fwd, xs).clone()
            ys.zero_()
            L.hopexp_run_win(R_, H_, g.fwd.ptr.data_ptr(), g.fwd.other.data_ptr(), g.fwd.w.data_ptr(), xs.data_ptr(),
                             xs.stride(0), ys.data_ptr(), ys.stride(0), n, current_stream_ptr(dev))
            torch.cuda.synchronize()
            okk = "exact" if torch.equal(ys, ref) else "MISMATCH"
            rot_win()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                rot_win()
            t = timeit(gr.replay, 20) / (3 * len(slabs))
            print(f"{name:5s} rotating slabs (> Infinity Cache), graph-replayed chains, windowed R={R_:3d} H={H_:2d} "
                  f"({(R_ + 2 * H_ + 1) // 2} KB LDS): {t * 1e3:7.2f} us per hop  frac {comp / t / 1e6 / 8000:.3f}  {okk}")
        for kind in ("plain", "rowmax", 0, 1, 24):
            rot(kind)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                rot(kind)
            t = timeit(gr.replay, 20) / (3 * len(slabs))
            label = kind if isinstance(kind, str) else f"v{kind:02d} {NAMES.get(kind, 'rec variant')}"
            print(f"{name:5s} rotating slabs (> Infinity Cache), graph-replayed chains, {label:32s}: {t * 1e3:7.2f} us per hop  frac {comp / t / 1e6 / 8000:.3f}")
        t = timeit(g.rebuild, 50)
        print(f"{name:5s} dc_graph_build (both sides, N={n} E={e}): {t * 1e3:7.2f} us")


if __name__ == "__main__":
    main()
