#!/bin/bash
# PMC passes over tools/pmc_hop.py (the step's F=256 hop launches): SQ issue / wait / LDS counters, L2 requests, HBM bytes.
# usage (on the GPU box, from the repo root): bash tools/exp/pmc_chain.sh gpurun_out/pmc_chain
set -u
OUT=${1:-gpurun_out/pmc_chain}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/tools/pmc_hop.py" > "$ROOT/$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run c TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
cd "$ROOT"
for p in a b c; do python3 tools/pmc_hop.py --parse-l2 "$OUT/$p" > "$OUT/$p.json" 2>"$OUT/$p.err"; done
python3 tools/pmc_hop.py --parse "$OUT/fetch" "$OUT/write" > "$OUT/traffic.json" 2>"$OUT/traffic.err"
# keep only the summaries (the raw csv trees are large)
for p in a b c fetch write; do rm -rf "$OUT/$p"; done
cat "$OUT"/a.json "$OUT"/b.json "$OUT"/c.json "$OUT"/traffic.json
