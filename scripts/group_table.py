"""Per-group budget table from a rocprofv3 --kernel-trace --stats summary (profiles/roundN_lrw_kernel_stats.csv).

    python scripts/group_table.py profiles/round4_lrw_kernel_stats.csv [steps]

Groups follow DESIGN.md section 3.  `steps` = profiled steps in the file (default: the call count of k_adamw).
"""
import csv
import re
import sys

GROUPS = [
    ("trunk weight gradients (k_igemm_wgrad_units / k_igemm_wgrad<128,2>, k_wgrad3x3_halo, short <64,3>)", r"k_igemm_wgrad_units|k_igemm_wgrad<|k_wgrad3x3_halo"),
    ("their fixed-order slab reductions (k_wgrad_unit_reduce, k_wgrad3_reduce)", r"k_wgrad_unit_reduce|k_wgrad3_reduce"),
    ("k_igemm_p8 (layer2/3/4 3x3 forward + data gradient, stride-2 forward)", r"k_igemm_p8"),
    ("BatchNorm apply / finalise passes", r"k_bn_"),
    ("fused encoder forward + backward (k_enc_fwd, k_enc_bwd)", r"k_enc_fwd|k_enc_bwd|k_enc_table|k_enc_bwd_table"),
    ("4-wave contraction, 64x64 tiles (encoder data gradients, heads)", r"k_igemm_fwd_glds<64"),
    ("4-wave contraction, 128-row tiles (stride-2 data gradients, 1x1 convolutions, heads)", r"k_igemm_fwd_glds<128"),
    ("stem (conv fwd, wgrad, BN+GELU+pool fwd, bwd, prep)", r"k_stem_"),
    ("layer1 k_conv3x3_c64", r"k_conv3x3_c64"),
    ("grouped weight gradients (encoder + heads)", r"k_igemm_wgrad_group|k_wgrad_group_table"),
    ("encoder row passes, attention backward, fused audio head (k_linear_ce), losses", r"k_add_ln|k_mha_|k_bias_act|k_embed|k_ce_|k_linear_ce|k_topk|k_avgpool|k_ls_|k_scale|k_lincomb"),
    ("fixed-order column sums (k_colsum)", r"k_colsum"),
    ("optimiser + shadows", r"k_adamw|k_grad_sumsq|k_transpose|k_opt_advance|k_cast_bf16"),
]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.reader(l for l in f if not l.startswith("#")):
            if r and r[0] != "Name":
                rows.append((r[0], int(r[1]), int(r[2])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else next(c for n, c, _ in rows if n.startswith("k_adamw"))
    tot = {g: [0, 0] for g, _ in GROUPS}
    other = [0, 0, []]
    for name, calls, ns in rows:
        for g, pat in GROUPS:
            if re.search(pat, name):
                tot[g][0] += calls
                tot[g][1] += ns
                break
        else:
            other[0] += calls
            other[1] += ns
            other[2].append((ns, name[:60]))
    print("| group | launches / step | us / step |")
    print("|---|---|---|")
    s_calls = s_ns = 0
    for g, _ in GROUPS:
        c, ns = tot[g]
        s_calls += c
        s_ns += ns
        print(f"| {g} | {c / steps:.0f} | {ns / steps / 1e3:,.0f} |")
    print(f"| other (fills, copies, torch elementwise) | {other[0] / steps:.0f} | {other[1] / steps / 1e3:,.0f} |")
    print(f"| **sum** | {(s_calls + other[0]) / steps:.0f} | {(s_ns + other[1]) / steps / 1e3:,.0f} |")
    for ns, n in sorted(other[2], reverse=True)[:8]:
        print(f"  other: {ns / steps / 1e3:7.1f} us  {n}", file=sys.stderr)


if __name__ == "__main__":
    main()
