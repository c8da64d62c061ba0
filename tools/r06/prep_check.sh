#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _ptr_array
dev = torch.device("cuda:0"); L = _lib.lib(); st = current_stream_ptr(dev)
fo = fi = 256; nseg = 4
ws = [torch.randn(fo, fi, device=dev) for _ in range(nseg)]
wmax = torch.empty(fo, device=dev); wimg = torch.empty(fo, nseg*fi, device=dev)
wt = torch.empty(fi, nseg*fo, device=dev); wtmax = torch.empty(fi, device=dev)
z = torch.empty(32768, device=dev)
def f(): L.dc_tag_weight_prep_zero(_ptr_array(ws), nseg, fo, fi, wmax.data_ptr(), wimg.data_ptr(), wt.data_ptr(), wtmax.data_ptr(), z.data_ptr(), z.numel(), current_stream_ptr(dev))
for _ in range(5): f()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"k_weight_prep (256 x 4 x 256, both images + zeroing 32768 floats): {e0.elapsed_time(e1) / 400 * 1e3:.2f} us per launch (graph replay)")
PY
true
