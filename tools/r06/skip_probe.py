#!/usr/bin/env python3
"""Round 6: what would the two-stream headline gain if a given launch cost NOTHING?  Timing only (results are wrong by
construction): the named C-ABI entries are replaced by no-ops, then bench.py's headline loop runs.
    python tools/r06/skip_probe.py <variant> [--serial-branches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deformcontact_amd import _lib  # noqa: E402

VARIANTS = {
    "base": [],
    "no_narrow_fwd": ["dc_tag_linear_fwd_split", "dc_tag_pack_weights"],
    "no_narrow_dw": ["dc_tag_linear_bwd_dw_split"],
    "no_narrow_hops": ["dc_spmm_f32_pack", "dc_spmm_f32"],
    "no_mask_grad": ["dc_tag_mask_grad"],
    "no_weight_prep": ["dc_tag_weight_prep_zero", "dc_tag_weight_prep"],
    "no_build": ["dc_graph_build_segmented"],
    "no_first_layers": ["dc_tag_linear_fwd_split", "dc_tag_pack_weights", "dc_tag_linear_bwd_dw_split", "dc_spmm_f32_pack",
                        "dc_spmm_f32"],
}


def main():
    variant = sys.argv[1]
    extra = sys.argv[2:]
    L = _lib.lib()
    for name in VARIANTS[variant]:
        getattr(L, name)                       # must exist
        setattr(L, name, lambda *a: 0)
    sys.argv = ["bench.py", "--headline-only", "--settle", "60", "--steps", "50", "--warmup", "5", "--windows", "2",
                "--no-pmc", "--no-kernel-trace"] + extra
    import bench
    print(f"== {variant} {' '.join(extra)}", flush=True)
    bench.main()


if __name__ == "__main__":
    main()
