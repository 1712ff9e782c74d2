#!/usr/bin/env python3
"""DESIGN.md section 3's kernel table from a rocprofv3 summary (tools/rocprof_summary.py output): launches and time PER STEP by kernel family, with the
bound and the algorithmic bytes per unit the design assigns to each family.   python tools/design_tables.py profiles/r05_kernel_stats_fp32x6.txt"""
import re
import sys

FAM = [  # (regex on the kernel name, family label, bound, algorithmic traffic per unit)
    (r"k_layer_x6<false, true", "256-wide layer, K = 3 input generated in-kernel (`clift_xyz_head_first2_x6_fwd`)", "bf16 MFMA + HBM write", "16 B in, 1 KB out per row (+ 32 B sign bytes)"),
    (r"k_layer_x6<false, false, 0", "256-wide layer, plain forward (`clift_gemm` precision 2)", "bf16 MFMA + HBM stream", "1 KB in, 1 KB out per row"),
    (r"k_layer_x6<false, false, [12]", "256-wide layer + E <= 4 output layer (`clift_xyz_head_last2_x6_fwd`)", "bf16 MFMA + HBM read", "1 KB in, 256 B of partial sums out (+ 1 KB when kept)"),
    (r"k_x6_out_sum", "... its 16-slot partial-sum fold", "HBM", "256 B in, 4 E B out per row"),
    (r"k_layer_x6<true, false, 0, false", "256-wide masked input gradient, mask = sign bytes", "bf16 MFMA + HBM stream", "1 KB + 32 B in, 1 KB out per row"),
    (r"k_layer_x6<true, false, 0, true", "second layer's input gradient consumed by the K = 3 weight gradient (`clift_xyz_head_first2_x6_bwd`)", "bf16 MFMA + HBM read", "1 KB + 16 B in per row"),
    (r"k_wgrad_x6<false>", "256 x 256 weight gradient (`clift_gemm` a_trans, precision 2)", "bf16 MFMA + 2 HBM streams", "2 KB in per row, 256 KB of atomics per block"),
    (r"k_wgrad_x6<true>", "... with X regenerated from the positions (`clift_xyz_head_first2_x6_wgrad`)", "bf16 MFMA + HBM read", "1 KB + 16 B in per row"),
    (r"k_layer_n6<10", "appearance layer 1 (160 -> 128), fp32x6 (`csrc/layer_n6.hip`)", "bf16 MFMA + HBM stream", "640 B in, 512 B out per row"),
    (r"k_layer_n6<8, 4, false, false, true", "appearance layer 2 + 3-wide output + sigmoid (`clift_app_head_last2_x6_fwd`)", "bf16 MFMA + HBM stream", "512 B in, 512 B (kept) + 12 B out per row"),
    (r"k_layer_n6<8, 4, true", "appearance masked input gradient 128 -> 128", "bf16 MFMA + HBM stream", "512 B + 512 B mask in, 512 B out per row"),
    (r"k_layer_n6<8, 5, true", "appearance input gradient 128 -> 160", "bf16 MFMA + HBM stream", "512 B in, 640 B out per row"),
    (r"k_wgrad_n6", "appearance weight gradients 128 x {128, 160}", "bf16 MFMA + 2 HBM streams", "1 - 1.1 KB in per row"),
    (r"k_app_front_fwd", "appearance front end: VM gather + basis Linear (fp32 MFMA) + input encoding (`clift_app_front_fwd_x`)", "L2 / LDS (table taps), VALU (sincos)", "16 B + 576 B of taps (L2) in, 640 B + 112 B (+ 576 B products when kept) out per row"),
    (r"k_app_gather_bwd", "appearance table scatter (`clift_app_gather_bwd`)", "L2 atomics (~28 %) + dependent LDS / memory round trips at 24 waves per CU", "576 B in per row, XCD-local atomics"),
    (r"k_density_bwd", "density table scatter (`clift_density_bwd`)", "L2 atomics (~28 %) + dependent LDS / memory round trips at 16 waves per CU", "8 B in per sample, XCD-local atomics"),
    (r"k_density_fwd", "density lookup + softplus (`clift_density_fwd`)", "L1 / L2 table reads", "4 B out per sample, 9 x 64 B taps (L2)"),
    (r"k_xcd_reduce", "fold of the eight XCD-private table-gradient copies", "HBM / L2", "8 x table size in"),
    (r"k_dgrad_narrow_stream<3", "semantic output layer backward: weight + masked input gradient in one pass (`clift_out_layer_bwd`)", "HBM stream", "1 KB + 96 B in, 1 KB out per row"),
    (r"k_dgrad_narrow_stream<1", "E <= 4 / 3-wide output layers' backward in one pass (`clift_out_layer_bwd[_nh]`)", "HBM stream", "hidden in, hidden-sized gradient out"),
    (r"k_dgrad_narrow_stream<4", "basis Linear input gradient 27 -> 144", "HBM stream", "112 B in, 576 B out per row"),
    (r"k_out_narrow_fwd", "semantic output layer + softmax (`clift_out_layer_fwd`)", "HBM read + fp32 MFMA", "1 KB in, 88 B out per row"),
    (r"k_wgrad_narrow_stream", "basis Linear weight gradient (`clift_wgrad_narrow`)", "HBM stream", "576 B + 112 B in per row"),
    (r"k_app_encode_bwd", "input-encoding backward", "HBM stream, VALU (sincos)", "640 B in, 112 B out per row"),
    (r"k_composite_bwd|k_composite_sum|k_composite_finish", "compositing forward / backward (+ the heads' output activations taken back)", "HBM stream, latency", "per-sample head outputs + weights"),
    (r"k_march|k_scan_counts|k_compact_fill|k_active_xyz", "ray marching (weights, prefix product, distortion loss), compaction", "latency / L2", "(N, S) sigma / alpha / T / w"),
    (r"k_adam|k_ema|k_tv_multi|k_pixel_losses|k_sf_|k_zero1|k_grad_shards|k_segment|k_contrastive|k_semantic_loss", "losses, TV, Adam, EMA, gradient shards", "latency (4 - 25 us launches)", "parameter-sized streams"),
    (r"at::native|__amd_rocclr", "torch glue (fills, copies, jitter, means)", "latency", "--"),
]

path = sys.argv[1]
rows = []
for l in open(path):
    m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", l)
    if m:
        rows.append((int(m.group(1)), float(m.group(2)), m.group(5).strip()))
steps = next(c for c, t, n in rows if n.startswith("k_pixel_losses"))
fam = {}
other = [0, 0.0]
for c, t, n in rows:
    for i, (rx, label, bound, traffic) in enumerate(FAM):
        if re.search(rx, n):
            f = fam.setdefault(i, [0, 0.0])
            f[0] += c
            f[1] += t
            break
    else:
        other[0] += c
        other[1] += t
total = sum(t for c, t, n in rows)
print(f"| kernel family (entry point) | launches / step | us / step | % of kernel time | bound by | algorithmic traffic |")
print("|---|---|---|---|---|---|")
for i in sorted(fam, key=lambda i: -fam[i][1]):
    c, t = fam[i]
    _, label, bound, traffic = FAM[i]
    print(f"| {label} | {c / steps:.1f} | {t / steps:.0f} | {100 * t / total:.1f} | {bound} | {traffic} |")
if other[0]:
    print(f"| (other) | {other[0] / steps:.1f} | {other[1] / steps:.0f} | {100 * other[1] / total:.1f} | | |")
print(f"| **all kernels** | {sum(c for c, t, n in rows) / steps:.0f} | **{total / steps:.0f}** | 100 | | ({steps} steps under rocprofv3, `{path}`) |")
