#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: the C-ABI library with dc_hopchain.hip compiled with extra defines
# (diagnostic builds; the other objects come from build/obj) -> build/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-unused-result "$@" \
  -c deformcontact_amd/csrc/dc_hopchain.hip -o build/variants/dc_hopchain_$name.o
objs=$(ls build/obj/*.o | grep -v dc_hopchain.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs build/variants/dc_hopchain_$name.o -o tools/r05/lib_$name.so
echo tools/r05/lib_$name.so
