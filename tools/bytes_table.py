"""HBM bytes per training step, tensor by tensor: what the DESIGN counts against what the PMC passes measured.

    python tools/bytes_table.py profiles/r06_pmc.json            # markdown table for DESIGN.md section 3

Left: every tensor the f16x3 step stores per SAMPLE (row of a field pass: 4096 rays x (64 coarse + 192 fine) = 1 048 576 rows per
step), who writes it, who reads it, bytes per row.  Right: the per-kernel FETCH_SIZE x 2 + WRITE_SIZE of the PMC run (gfx950
correction of MI355X_MICROARCH.md) x launches per step.  "Required" = operands of a weight gradient (written once by a field
kernel, read once by a weight-gradient kernel): the floor of fp32-accurate stored operands."""
import json
import sys

ROWS = 4096 * (64 + 64 + 128)  # field evaluations per step (coarse pass 64, fine pass 192 samples per ray)

# tensor, bytes per row, written by, read by (kernel class x times), required (weight-gradient operand)?
TENSORS = [
    ("x0 (masked encoding, 63 -> 64 floats)", 256, "field_fwd", [("field_fwd (skip layer, L2)", 1), ("field_bwd", 1), ("wgrad", 2)], "operand of two weight gradients (layer 0, skip layer)"),
    ("h_0 .. h_7 (post-ReLU trunk activations, fp32)", 8 * 1024, "field_fwd", [("wgrad", 1)], "REQUIRED: B operand of the next layer's weight gradient"),
    ("e (xyz_encoding_final output, fp32)", 1024, "field_fwd", [("composite_fwd", 1), ("composite_bwd", 1), ("wgrad", 1)], "operand of the head layers' weight gradient; the two compositing reads are the removable part (see text)"),
    ("g1, g2, r1 (128-wide head activations)", 3 * 512, "field_fwd", [("field_bwd: g2, r1", 2 / 3), ("composite_fwd, composite_bwd: g2", 2 / 3), ("wgrad: g1", 1 / 3)], "ReLU masks + vector heads' column sums (backward), candidate compositing, candidate weight gradient"),
    ("sign bits (9 slots x 8 B per lane)", 9 * 32, "field_fwd", [("field_bwd", 1)], "instead of re-reading 9 activations in the backward pass"),
    ("gz_0 .. gz_7 (pre-activation gradients, fp32)", 8 * 1024, "field_bwd", [("wgrad", 1)], "REQUIRED: A operand of the layer's weight gradient"),
    ("gz_e", 1024, "field_bwd", [("wgrad", 1)], "REQUIRED (final layer)"),
    ("gz_g2, [gz_r1 | gz_g1]", 3 * 512, "field_bwd", [("wgrad", 1)], "REQUIRED (head layers)"),
    ("per-sample scalars (sigma, rgb, weights, d sigma, d pre, d xyz ...)", 120, "field / composite kernels", [("field / composite kernels", 1)], ""),
]


def main():
    pmc = json.load(open(sys.argv[1]))
    fwd = next(k for k in pmc if k.startswith("field16_fwd_kernel<2"))
    steps = pmc[fwd]["dispatches"] / 2
    print("| tensor | B / row written | B / row read (by) | GB / step | role |")
    print("|---|---|---|---|---|")
    tot_w = tot_r = req = 0.0
    for name, b, wr, readers, role in TENSORS:
        rd = sum(b * n for _, n in readers)
        gb = (b + rd) * ROWS / 1e9
        tot_w += b
        tot_r += rd
        if role.startswith("REQUIRED"):
            req += gb
        print(f"| {name} | {b} ({wr}) | {rd:.0f} ({'; '.join(r for r, _ in readers)}) | {gb:.2f} | {role} |")
    print(f"| **sum of the count** | {tot_w:.0f} | {tot_r:.0f} | **{(tot_w + tot_r) * ROWS / 1e9:.1f}** | of which REQUIRED (fp32 operands written once, read once): **{req:.1f}** |")
    print()
    groups = {"field_fwd": ["field16_fwd_kernel"], "field_bwd": ["field16_bwd_kernel"], "weight gradients (f16x3, all shapes)": ["wgrad_f16x3_kernel"],
              "compositing": ["composite_fwd_kernel", "composite_bwd_kernel"]}
    counted = {"field_fwd": (256 + 8192 + 1024 + 1536 + 288 + 30) * ROWS + 256 * ROWS, "field_bwd": (8192 + 1024 + 1536 + 60) * ROWS + (1024 + 288 + 256 + 40) * ROWS,
               "weight gradients (f16x3, all shapes)": (2 * 8192 + 2 * 256 + 2 * 1024 + 2 * 1024 + 1024) * ROWS, "compositing": 2 * (1024 + 512 + 40) * ROWS}
    print("| kernel class | launches / step | measured fetch GB | measured write GB | measured sum | counted | measured / counted |")
    print("|---|---|---|---|---|---|---|")
    total = 0.0
    seen = set()
    for g, pats in groups.items():
        f = w = n = 0.0
        for k, v in pmc.items():
            if isinstance(v, dict) and any(k.startswith(p) for p in pats):
                seen.add(k)
                n += v["dispatches"] / steps
                f += v.get("fetch_bytes_per_launch", 0) * v["dispatches"] / steps
                w += v.get("write_bytes_per_launch", 0) * v["dispatches"] / steps
        total += f + w
        print(f"| {g} | {n:.0f} | {f / 1e9:.2f} | {w / 1e9:.2f} | {(f + w) / 1e9:.2f} | {counted[g] / 1e9:.2f} | {(f + w) / counted[g]:.2f} |")
    rest = sum((v.get("fetch_bytes_per_launch", 0) + v.get("write_bytes_per_launch", 0)) * v["dispatches"] / steps
               for k, v in pmc.items() if isinstance(v, dict) and "dispatches" in v and k not in seen)
    total += rest
    print(f"| every other kernel of the step | | | | {rest / 1e9:.2f} | | |")
    print(f"| **step** | | | | **{total / 1e9:.1f}** | **{sum(counted.values()) / 1e9:.1f}** | {total / sum(counted.values()):.2f} |")


if __name__ == "__main__":
    main()
