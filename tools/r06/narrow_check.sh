#!/bin/bash
# (the DC_NARROW_MB switch these lines set existed while the tile height was being chosen: the library now always takes 32-row tiles)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_narrow
mkdir -p $O
timeout 600 python -m pytest tests/test_narrow_dense.py tests/test_full_size.py -x -q -m gpu 2>&1 | tail -n 6
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_narrow/narrow_time.txt
import torch, os
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array
dev = torch.device("cuda:0"); L = _lib.lib(); st = current_stream_ptr(dev)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    ts=[]
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/reps*1e3)
    return sorted(ts)[2]
for name, n, fi, wpad in (("soft", 32768, 21, 96), ("rigid", 24384, 25, 112)):
    fo = 256
    slabs = [torch.randn(n, wpad, device=dev) for _ in range(6)]
    outs = [ops._alloc_slab(n, 1024, dev)[:, :fo] for _ in range(6)]      # outputs inside the next layer's slab, as in a step
    ws = [torch.randn(fo, fi, device=dev)/fi**.5 for _ in range(4)]
    bias = torch.randn(fo, device=dev)
    wcat = torch.empty(fo, wpad, device=dev)
    i = [0]
    def new():
        i[0] += 1; s, o = slabs[i[0] % 6], outs[i[0] % 6]
        L.dc_tag_linear_fwd_narrow(s.data_ptr(), s.stride(0), _ptr_array(ws), 4, fi, bias.data_ptr(), 1, o.data_ptr(), o.stride(0), n, wpad, fo, st)
    def old():
        i[0] += 1; s, o = slabs[i[0] % 6], outs[i[0] % 6]
        L.dc_tag_pack_weights(_ptr_array(ws), 4, wcat.data_ptr(), fo, fi, wpad, st)
        L.dc_tag_linear_fwd_split(_ptr_array([s]), _i64_array([s.stride(0)]), _ptr_array([wcat]), 1, bias.data_ptr(), 1, o.data_ptr(), o.stride(0), n, wpad, fo, 6, st)
    mb = n * (wpad + fo) * 4 / 1e6
    for tag, fn in (("k_fwd_narrow", new), ("pack + k_fwd_split", old), ("k_fwd_narrow", new)):
        t = timeit(fn)
        print(f"{name:5s} {tag:20s} {t:6.1f} us  {mb / t * 1e-3 * 1e3:7.1f} GB/s = {mb / t / 8e3 * 1e3:.3f} of 8 TB/s", flush=True)
PY
for mb in 1 2; do echo "DC_NARROW_MB=$mb"; DC_NARROW_MB=$mb python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_narrow/narrow_time.txt
import torch
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _ptr_array
dev = torch.device("cuda:0"); L = _lib.lib(); st = current_stream_ptr(dev)
for name, n, fi, wpad in (("soft", 32768, 21, 96), ("rigid", 24384, 25, 112)):
    fo = 256
    slabs = [torch.randn(n, wpad, device=dev) for _ in range(6)]
    outs = [ops._alloc_slab(n, 1024, dev)[:, :fo] for _ in range(6)]
    ws = [torch.randn(fo, fi, device=dev)/fi**.5 for _ in range(4)]
    bias = torch.randn(fo, device=dev)
    def new(k):
        s, o = slabs[k % 6], outs[k % 6]
        L.dc_tag_linear_fwd_narrow(s.data_ptr(), s.stride(0), _ptr_array(ws), 4, fi, bias.data_ptr(), 1, o.data_ptr(), o.stride(0), n, wpad, fo, st)
    for k in range(5): new(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(100): new(k)
    e1.record(); torch.cuda.synchronize()
    print(f"  {name}: {e0.elapsed_time(e1) * 10:.1f} us")
PY
done
timeout 200 python tools/r06/skip_probe.py base 2>&1 | grep -E "^==|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  two streams: \1 ms/'
DC_NARROW_FWD=0 timeout 200 python tools/r06/skip_probe.py base 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  two streams, DC_NARROW_FWD=0: \1 ms/'
timeout 200 python tools/r06/skip_probe.py base --serial-branches 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  one stream: \1 ms/'
DC_NARROW_FWD=0 timeout 200 python tools/r06/skip_probe.py base --serial-branches 2>&1 | grep -E "ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/  one stream, DC_NARROW_FWD=0: \1 ms/'
