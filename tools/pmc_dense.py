#!/usr/bin/env python3
"""PMC summary of the layer-2 dense blocks (the kernels `roofline_mfma` prices), for profiles/.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
        --output-format csv -d out/a -- python3 tools/pmc_dense.py
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA \
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
        --output-format csv -d out/b -- python3 tools/pmc_dense.py
    python tools/pmc_dense.py --parse out/a out/b > profiles/r02/dense_pmc.json

(separate passes: 8 SQ slots per pass; no tracing domain besides --kernel-trace next to --pmc)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench                                          # the very launches roofline_mfma times
    bench.dense_roofline(torch.device("cuda:0"), 32768, 24384, 4)
    torch.cuda.synchronize()


def parse(dirs):
    per = {}
    for d in dirs:
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(t in name for t in ("k_fwd_h2", "k_dw_h2w", "k_dw_split", "k_fwd_split", "k_fwd_bf16")):
                continue
            key = (name.split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
            e = per.setdefault(key, {})
            c = e.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"])
            c[1] += 1
            dur = e.setdefault("_dur_ns", [0.0, 0])
            dur[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dur[1] += 1
    out = []
    for (name, grid), e in sorted(per.items()):
        avg = {k: v[0] / v[1] for k, v in e.items()}
        row = {"kernel": name, "grid_threads": grid, "avg_duration_us_profiled": round(avg.pop("_dur_ns") / 1e3, 1),
               "counters_avg_per_launch": {k: round(v, 1) for k, v in sorted(avg.items())}}
        g = row["counters_avg_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g and g["GRBM_GUI_ACTIVE"] > 0:
            # MFMA pipe busy cycles summed over the 1024 SIMDs / (shader cycles of the launch x 1024);
            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS section)
            row["mfma_pipe_busy_frac"] = round(g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
            if g.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None and g["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
                row["valu_mfma_coexec_frac_of_mfma_busy"] = round(
                    g["SQ_VALU_MFMA_COEXEC_CYCLES"] / g["SQ_VALU_MFMA_BUSY_CYCLES"], 4)
        if "SQ_LDS_BANK_CONFLICT" in g and g.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            row["lds_bank_conflict_frac_of_lds_active"] = round(g["SQ_LDS_BANK_CONFLICT"] / g["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_WAVE_CYCLES" in g and g["SQ_WAVE_CYCLES"] > 0:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in g:
                    row[k.lower() + "_frac_of_wave_cycles"] = round(g[k] / g["SQ_WAVE_CYCLES"], 4)
        if "SQ_INSTS_VALU" in g and g.get("SQ_INSTS_MFMA", 0) > 0:
            row["valu_per_mfma"] = round(g["SQ_INSTS_VALU"] / g["SQ_INSTS_MFMA"], 2)
        out.append(row)
    print(json.dumps({"note": "layer-2 dense blocks (soft N=32768, rigid N=24384; K=1024, Fo=256), averages per launch "
                              "over the launches of bench.dense_roofline; profiled runs clock lower than unprofiled "
                              "ones - never compare these durations with bench.py's", "kernels": out}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--parse":
        parse(sys.argv[2:])
    else:
        run()
