#!/usr/bin/env python3
"""Step budget by kernel family from a tools/rocprof_summary.py table:  python tools/step_budget.py <kernel_stats.txt> <steps>"""
import re
import sys

FAMILIES = [
    ("256 x 256 forward layers (`k_layer_x6<false,*>` + `k_x6_out_sum`)", r"k_layer_x6<false|k_x6_out_sum|k_layer_f32<false|k_head_bf16_fwd|k_layer_bf16<false"),
    ("256 x 256 weight gradients (`k_wgrad_x6<*>`)", r"k_wgrad_x6|k_wgrad_bf16|k_wgrad_n128_stream<32, true|k_wgrad_f32"),
    ("256 x 256 input gradients (`k_layer_x6<true,*>`)", r"k_layer_x6<true|k_layer_f32<true|k_layer_bf16<true"),
    ("narrow layers (`k_*_narrow_stream`, `k_out_narrow_fwd`, `k_gemm<256,32>`, `k_rows_act_*`)", r"narrow|k_gemm|k_rows_act|k_linear_k3"),
    ("128-wide appearance layers (`k_layer_n128`, `k_wgrad_n128_stream`; exact fp32)", r"k_layer_n128|k_wgrad_n128"),
    ("appearance front end + input-assembly backward (`k_app_front_fwd`, `k_app_encode_*`)", r"k_app_front|k_app_encode|k_app_gather_fwd"),
    ("VM-table density lookup / scatters (`k_density_*`, `k_app_gather_bwd_u`, `k_xcd_reduce`)", r"k_density|k_app_gather_bwd|k_xcd_reduce"),
    ("march / composite / compaction", r"k_march|k_composite|k_scan|k_compact|k_active_xyz"),
    ("losses, TV, Adam, EMA, gradient shards", r"k_pixel|k_tv|k_adam|k_ema|k_sf_|k_grad_shards|k_zero1|k_segment|k_contrastive"),
    ("torch fills / copies / random numbers / reductions", r"at::native|rocclr"),
]


def main():
    steps = float(sys.argv[2])
    fam = {f[0]: [0, 0.0] for f in FAMILIES}
    other, total = [0, 0.0], 0.0
    for l in open(sys.argv[1]):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", l)
        if not m or "TOTAL" in l:
            continue
        calls, tot, name = int(m.group(1)), float(m.group(2)), m.group(5)
        total += tot
        for label, pat in FAMILIES:
            if re.search(pat, name):
                fam[label][0] += calls; fam[label][1] += tot
                break
        else:
            other[0] += calls; other[1] += tot
            print("unassigned:", name[:80], file=sys.stderr)
    print("| family | launches | us / step | share |\n|---|---|---|---|")
    for label, _ in FAMILIES:
        c, t = fam[label]
        print(f"| {label} | {c / steps:.0f} | {t / steps:.0f} | {100 * t / total:.1f} % |")
    if other[1]:
        print(f"| other | {other[0] / steps:.0f} | {other[1] / steps:.0f} | {100 * other[1] / total:.1f} % |")
    print(f"total {total / steps:.0f} us per step")


if __name__ == "__main__":
    main()
