import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from deformcontact_amd import _lib
from deformcontact_amd.graph import current_stream_ptr
from deformcontact_amd.ops import _i64_array, _ptr_array
from kbench import timeit
dev = torch.device("cuda:0"); L = _lib.lib(); st = current_stream_ptr(dev)
fi, nseg, fo = 256, 4, 256
for n in (64, 64*32, 64*128, 64*256, 64*384, 64*512, 64*640, 64*768, 64*1024):
    slab = torch.randn(n, nseg*fi, device=dev)
    xs = [slab[:, s*fi:(s+1)*fi] for s in range(nseg)]; ld = [nseg*fi]*nseg
    ws = [torch.randn(fo, fi, device=dev)/16 for _ in range(nseg)]
    bias = torch.randn(fo, device=dev); out = torch.empty(n, fo, device=dev)
    pa_x, pa_w, pa_ld = _ptr_array(xs), _ptr_array(ws), _i64_array(ld)
    ms = timeit(lambda: L.dc_tag_linear_fwd(pa_x, pa_ld, pa_w, nseg, bias.data_ptr(), 1, out.data_ptr(), fo, n, fi, fo, st), 20)
    blocks = (n // 64) * 2
    print(f"N={n:6d} blocks={blocks:5d} ({blocks/256:.2f}/CU): {ms*1e3:8.1f} us  {2.0*n*fi*nseg*fo/ms/1e9:7.1f} TF/s")
