#!/bin/bash
# round 6: where do the microseconds of the first layers' dW block (k_dw_split<2, true, 6>) go?  timing-only builds
cd "$(dirname "$0")/../.."
if [ "$1" = "build" ]; then
  python -m deformcontact_amd.build > /dev/null
  mkdir -p build/variants
  for b in 1 2 4 8 16 31; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-unused-result -DDC_DWS_ABL=$b \
      -c deformcontact_amd/csrc/dc_dense_split.hip -o build/variants/dc_dense_split_abl$b.o &
  done
  wait
  for b in 1 2 4 8 16 31; do
    objs=$(ls build/obj/*.o | grep -v dc_dense_split.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs build/variants/dc_dense_split_abl$b.o -o tools/r06/lib_dws_abl$b.so
  done
  ls tools/r06/lib_dws_abl*.so
  exit 0
fi
O=gpurun_out/r06_dw; mkdir -p $O
{
echo "== product"; python tools/r06/dw_narrow_time.py 2>&1 | grep -v amdgpu.ids
for b in 1 2 4 8 16 31; do
  echo "== ablation $b (1 no MFMA, 2 no mask loads, 4 no partial stores, 8 no gradient loads, 16 no x loads)"
  python tools/exp/run_with_lib.py tools/r06/lib_dws_abl$b.so tools/r06/dw_narrow_time.py 2>&1 | grep -v amdgpu.ids
done
} | tee $O/dw_abl.txt
