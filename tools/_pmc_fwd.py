import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deformcontact_amd import _lib, ops
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array
dev = torch.device("cuda:0"); L = _lib.lib(); st = current_stream_ptr(dev)
n, fi, nseg, fo = 32768, 256, 4, 256
slab = torch.randn(n, nseg*fi, device=dev)
xs = [slab[:, s*fi:(s+1)*fi] for s in range(nseg)]; ld = [nseg*fi]*nseg
ws = [torch.randn(fo, fi, device=dev)/16 for _ in range(nseg)]
bias = torch.randn(fo, device=dev); out = torch.empty(n, fo, device=dev)
rowmax = slab.abs().amax(1).contiguous(); wmax = ops.weight_rowmax(ws)
pa_x, pa_w, pa_ld = _ptr_array(xs), _ptr_array(ws), _i64_array(ld)
for _ in range(5):
    L.dc_tag_linear_fwd_h2(pa_x, pa_ld, pa_w, nseg, bias.data_ptr(), 1, out.data_ptr(), fo, n, fi, fo, rowmax.data_ptr(), wmax.data_ptr(), st)
torch.cuda.synchronize()
