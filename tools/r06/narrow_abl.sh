#!/bin/bash
# (the DC_NARROW_MB switch these lines set existed while the tile height was being chosen: the library now always takes 32-row tiles)
# round 6: where do k_fwd_narrow's microseconds go?  timing-only builds of dc_dense_narrow.hip (-DDC_NARROW_ABL=<bits>)
cd "$(dirname "$0")/../.."
if [ "$1" = "build" ]; then
  python -m deformcontact_amd.build > /dev/null
  mkdir -p build/variants
  for b in 1 2 4 8 16 31; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-unused-result -DDC_NARROW_ABL=$b \
      -c deformcontact_amd/csrc/dc_dense_narrow.hip -o build/variants/dc_dense_narrow_abl$b.o
    objs=$(ls build/obj/*.o | grep -v dc_dense_narrow.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs build/variants/dc_dense_narrow_abl$b.o -o tools/r06/lib_narrow_abl$b.so
  done
  ls tools/r06/lib_narrow_abl*.so
  exit 0
fi
O=gpurun_out/r06_narrow
mkdir -p $O
cat > /tmp/nt.py <<'PY'
import torch, sys
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
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(100): new(k)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 10)
    print(f"  {name}: {sorted(ts)[1]:.1f} us", flush=True)
PY
for mb in 1 2; do
  echo "== DC_NARROW_MB=$mb product"; DC_NARROW_MB=$mb python /tmp/nt.py 2>&1 | grep -v amdgpu.ids
  for b in 1 2 4 8 16 31; do
    echo "== DC_NARROW_MB=$mb ablation $b (1 no MFMA, 2 no weight image, 4 no row stores, 8 no row loads / planes, 16 no staging)"
    DC_NARROW_MB=$mb python tools/exp/run_with_lib.py tools/r06/lib_narrow_abl$b.so /tmp/nt.py 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/narrow_abl.txt
