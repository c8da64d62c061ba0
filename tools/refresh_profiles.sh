#!/bin/bash
# Everything profiles/rNN/ cites, in one pass on a GPU box:   tools/refresh_profiles.sh gpurun_out/<tag>
# (pytest -m gpu, smoke, the default bench line incl. its own --pmc child passes, kbench, rocprofv3 kernel stats of
# the step with serial and with two-stream branches, the two PMC passes over the layer-2 dense blocks).
set -u
if [ $# -lt 1 ] || [ -z "$1" ] || [ "${1#/}" != "$1" ]; then
    echo "usage: tools/refresh_profiles.sh <output dir RELATIVE to the repo root, e.g. gpurun_out/p>" >&2
    exit 2
fi
O=$1
R=$(pwd)
mkdir -p "$R/$O"
export TMPDIR=/tmp
cd /tmp
(cd "$R" && timeout -s KILL 1200 python -m pytest tests -m gpu -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt")
(cd "$R" && timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1; tail -2 "$O/smoke.txt")
(cd "$R" && S=$(date +%s) && timeout -s KILL 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench wall $(( $(date +%s) - S )) s" | tee "$O/bench_wall.txt"; tail -c 300 "$O/bench.json")
(cd "$R" && timeout -s KILL 300 python tools/kbench.py > "$O/kbench.txt" 2>&1)
Q="--no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --no-backbones"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_serial" -- python3 "$R/bench.py" $Q --serial-branches > "$R/$O/stats_serial.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_default" -- python3 "$R/bench.py" $Q > "$R/$O/stats_default.log" 2>&1
# the GCNConv / GATConv encoders on the same batch (dc_gat_*, dc_sddmm_f32, self-loop adjacency build)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_backbones" -- python3 "$R/bench.py" --no-cpu-baseline --no-full-step --no-strict-fp32 --no-radius100k --no-pmc --no-merged --steps 5 --warmup 2 > "$R/$O/stats_backbones.log" 2>&1
# the whole reference step at B = 32 (attention kernels), eager
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_full_b32" -- python3 "$R/tools/full_step.py" --batch 32 --steps 3 > "$R/$O/stats_full_b32.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/pmc_a" -- python3 "$R/tools/pmc_dense.py" > "$R/$O/pmc_a.log" 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_b" -- python3 "$R/tools/pmc_dense.py" > "$R/$O/pmc_b.log" 2>&1
(cd "$R" && python tools/pmc_dense.py --parse "$O/pmc_a" "$O/pmc_b" > "$O/dense_pmc.json" 2> "$O/dense_pmc.err")
# L1 -> L2 and L2 -> fabric read requests of the F=256 hop: per-branch launches and ONE launch over the merged adjacency
for M in 0 1; do
    DC_MERGE_BRANCHES=$M timeout -s KILL 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/$O/pmc_hop_l2_$M" -- python3 "$R/tools/pmc_hop.py" > "$R/$O/pmc_hop_l2_$M.log" 2>&1
    (cd "$R" && python tools/pmc_hop.py --parse-l2 "$O/pmc_hop_l2_$M" > "$O/hop_l2_requests_merged$M.json" 2>> "$O/dense_pmc.err")
done
# r04: the 3-hop chain launches (dc_hop_chain_f32): timing against 12 single hops, SQ / L2 / HBM counters, ablations
(cd "$R" && timeout -s KILL 300 python tools/exp/hop_chain.py > "$O/hop_chain_timing.txt" 2>&1)
(cd "$R" && timeout -s KILL 900 bash tools/exp/pmc_chain.sh "$O/pmc_chain" > "$O/pmc_chain.log" 2>&1)
(cd "$R" && timeout -s KILL 600 python tools/exp/hop_chain_abl.py 0 1 2 3 > "$O/hop_chain_ablation.txt" 2>&1)
(cd "$R" && cp gpurun_out/parity_distances.json "$O/parity_distances.json" 2>/dev/null)
(cd "$R" && timeout -s KILL 300 python tools/exp/build_time.py > "$O/graph_build_time.txt" 2>&1)
(cd "$R" && DC_BF16_X=0 timeout -s KILL 300 python tools/exp/bf16_dense.py > "$O/bf16_dense_128tiles.txt" 2>&1; DC_BF16_X=1 timeout -s KILL 300 python tools/exp/bf16_dense.py > "$O/bf16_dense_256tiles.txt" 2>&1)
(cd "$R" && timeout -s KILL 300 python tools/exp/gap_parts.py > "$O/new_batch_gap_parts.txt" 2>&1)
(cd "$R" && timeout -s KILL 300 python tools/exp/attn_flash.py > "$O/attn_flash_fwd.txt" 2>&1; timeout -s KILL 300 python tools/exp/attn_flash_bwd.py > "$O/attn_flash_bwd.txt" 2>&1)
(cd "$R" && timeout -s KILL 300 python tools/exp/hop_colsplit.py > "$O/hop_column_split.txt" 2>&1)
(cd "$R" && timeout -s KILL 300 python tools/exp/attn_flash_stress.py 40 > "$O/attn_flash_stress.txt" 2>&1)
(cd "$R" && for b in 32 16; do timeout -s KILL 300 python tools/full_step.py --batch $b --steps 5 --graph 2>&1 | tail -1; done > "$O/full_step_graph.txt"; timeout -s KILL 300 python tools/full_step.py --batch 32 --steps 3 2>&1 | tail -1 >> "$O/full_step_graph.txt")
# r04: the forward-shaped dense block with both operands by LDS-DMA (k_fwd_h2d) against k_fwd_h2w, its ablations and its
# launch anatomy; the B = 4 step with stock and with library attention; same-box A/B of the two r04 switches
(cd "$R" && H2S_SRC=dense_h2d.hip timeout -s KILL 400 python tools/exp/dense_h2s.py 0 34 2 32 > "$O/dense_h2d_timing.txt" 2>&1)
(cd "$R" && timeout -s KILL 300 python tools/exp/dense_h2d_trace.py > "$O/dense_h2d_anatomy.txt" 2>&1)
(cd "$R" && for f in 0 1; do echo "DC_FUSED_ATTN=$f"; DC_FUSED_ATTN=$f timeout -s KILL 300 python tools/full_step.py --batch 4 --steps 50 --graph 2>&1 | tail -1; done > "$O/attn_b4_stock_vs_library.txt")
(cd "$R" && timeout -s KILL 600 tools/exp/ab_headline.sh DC_H2_DMA=0 DC_H2_DMA=1 > "$O/ab_headline.txt" 2>&1; timeout -s KILL 600 tools/exp/ab_headline.sh DC_HOP_CHAIN=0 DC_HOP_CHAIN=1 >> "$O/ab_headline.txt" 2>&1)
# r04 (last day): the global adjacency build of the relabelled 100k-point radius graph, windowed vs bucketed, per kernel;
# the same on ordered mesh batches; the kernel list of configs[4]'s forward + backward with and without the LDS-DMA dW block
(cd "$R" && timeout -s KILL 200 python tools/exp/prep_parts.py > "$O/prep_parts.txt" 2>&1)
(cd "$R" && GRAFT_REPO_ROOT="$R" timeout -s KILL 300 bash tools/exp/csr_ab.sh DC_CSR_BUCKETS=0 DC_CSR_BUCKETS=1 > "$O/csr_build_windowed_vs_bucketed.txt" 2>&1)
(cd "$R" && for b in 32 128; do for m in 0 1; do echo "== B=$b DC_CSR_BUCKETS=$m"; DC_CSR_BUCKETS=$m timeout -s KILL 200 python tools/exp/build_time.py $b 2>&1 | grep "us per build"; done; done > "$O/csr_build_mesh_batches.txt")
for M in 0 1; do
    (cd /tmp && DC_DW_BF16_DMA=$M timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/radius100k_dwdma$M" -- python3 "$R/bench.py" --workload radius100k --no-cpu-baseline > "$R/$O/radius100k_dwdma$M.log" 2>&1)
done
# keep the summaries, drop the bulky per-dispatch traces (gpurun_out/ travels back, 64 MiB cap)
find "$R/$O" -name "*kernel_trace.csv" -delete
find "$R/$O" -name "*counter_collection.csv" -delete
ls -la "$R/$O"
