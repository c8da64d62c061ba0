#!/bin/bash
# round 6, VERDICT r05 item 7: diagnostic builds of dc_hopchain.hip for the chain-miscompute hunt, all with the TIGHT LDS
# request of rounds 1-4 (the condition under which 35 of 2,000 two-stream train steps differed):
#   tight      control (LDS-DMA staging, tight request)
#   poison     every byte of the workgroup's LDS starts as a quiet NaN: a NaN in the result = a read of LDS before its data landed
#   regstage   the slice goes global -> registers -> ds_write instead of LDS-DMA
#   readback   LDS-DMA, and every DMA wave reads back its last piece behind its own vmcnt(0), before the barrier
set -e
cd "$(dirname "$0")/../.."
python -m deformcontact_amd.build > /dev/null
bash tools/r05/build_variant.sh r06_tight -DDC_CHAIN_LDS_TIGHT
bash tools/r05/build_variant.sh r06_poison -DDC_CHAIN_LDS_TIGHT -DDC_CHAIN_POISON
bash tools/r05/build_variant.sh r06_regstage -DDC_CHAIN_LDS_TIGHT -DDC_CHAIN_REGSTAGE
bash tools/r05/build_variant.sh r06_readback -DDC_CHAIN_LDS_TIGHT -DDC_CHAIN_DMA_READBACK
mv tools/r05/lib_r06_*.so tools/r06/
ls -la tools/r06/*.so
