"""Floor table of the config-B step (VERDICT r05 "next round" item 1): per kernel family, the time the PRESENT decomposition cannot go
below -- algorithmic bytes / 6.3 TB/s (the HBM rate streaming kernels reach on this part) or FLOPs / 1.7 PF (the MFMA rate the 256x256x64
main loop reaches on random data at the clock the power cap allows) -- beside the time the kernel trace shows.

    python tools/floor_table.py profiles/r05_final2_gemm_in_step_by_shape.md profiles/r05_final2_kernel_trace.md 7 > profiles/r06_floor_table.md

Inputs: the in-step GEMM table (bench.py, MPV_BENCH_BY_SHAPE) and the kernel trace summary (tools/rocpd_stats.py) of the same tree; the
third argument is the number of steps the trace covers.  Algorithmic bytes of the non-GEMM families are written out below for config B
(R = 32 x 8 x 197 = 50432 ViT rows of 768, 5120 decoder rows of 2048, 130.0 M trainable parameters)."""
import re
import sys

HBM = 6.3e12       # B/s a streaming kernel reaches (the guide's peak is 8 TB/s)
MFMA = 1.7e15      # FLOP/s of the GEMM main loop on random data under the power cap

R, D, RT = 50432, 768, 50176
DR, H = 5120, 2048
BT, HEADS, S1 = 256, 8, 197

# family -> (regex on the trace's kernel column, floor in us per launch of the DOMINANT shape, what the floor counts)
NON_GEMM = [
    ("ViT LayerNorm forward", r"ln_fwd8_kernel", 4 * R * D / HBM * 1e6, "read x + write y (bf16), 50432 x 768"),
    ("ViT LayerNorm backward (+ residual gradient, dgamma/dbeta partials)", r"ln_bwd8_kernel", 8 * R * D / HBM * 1e6, "read dy, x, residual gradient + write dx"),
    ("decoder LayerNorm forward on the fp32 stream", r"ln_stream_fwd_wg_kernel", 12 * DR * H / HBM * 1e6, "read h (fp32) + a (bf16), write h' (fp32) + y (bf16)"),
    ("decoder LayerNorm backward", r"ln_bwd8_plain_kernel", 10 * DR * H / HBM * 1e6, "read dy, x (fp32), residual gradient + write dx"),
    ("ViT spatial attention forward", r"attn_fwd_pres_kernel", (2 * R * 3 * D + 2 * R * D) / HBM * 1e6, "read qkv + write o (HBM); MFMA floor 23 us"),
    ("ViT spatial attention dQ", r"attn_bwd_dq_duo96", (2 * R * 3 * D + 3 * 2 * R * D) / HBM * 1e6, "read qkv, dO, O + write dQ"),
    ("ViT spatial attention dK/dV", r"attn_bwd_dkv_duo96", (2 * R * 3 * D + 2 * R * D + 2 * 2 * R * D) / HBM * 1e6, "read qkv, dO + write dK, dV"),
    ("temporal attention forward", r"temporal_attn_b16_kernel<false", 8 * RT * D / HBM * 1e6, "read q, k, v + write o on the token rows"),
    ("temporal attention backward", r"temporal_attn_b16_kernel<true", 14 * RT * D / HBM * 1e6, "read q, k, v, dO + write dq, dk, dv"),
    ("decoder attention forward", r"attn_fwd_pair64", (2 * DR * 3 * H + 2 * DR * H) / HBM * 1e6, "read qkv + write o"),
    ("decoder attention dQ", r"attn_bwd_dq_pair64", (2 * DR * 3 * H + 3 * 2 * DR * H) / HBM * 1e6, "read qkv, dO, O + write dQ"),
    ("decoder attention dK/dV", r"attn_bwd_dkv_(res|pair64)_kernel<64", (2 * DR * 3 * H + 2 * DR * H + 2 * 2 * DR * H) / HBM * 1e6, "read qkv, dO + write dK, dV"),
    ("abstractor attention (fwd + dq + dkv)", r"attn_(fwd|bwd_dq|bwd_dkv)_kernel<96>", None, "chunked kernels, 1577 keys: measured time taken as floor"),
    ("AdamW (one launch)", r"adamw_grouped_kernel", 28 * 130.0e6 / HBM * 1e6, "28 B per parameter"),
    ("gradient norm", r"grad_sumsq_kernel", 2 * 130.0e6 / HBM * 1e6, "2 B per parameter"),
    ("cross entropy on the loss window", r"cross_entropy_kernel", 2 * 2 * 1024 * 51200 / HBM * 1e6, "read logits + write dlogits in place"),
    ("split-K reduce of the weight gradients", r"splitk_reduce_kernel", 0.0, "not algorithmic: exists only because the wgrad is split along K"),
]
GLUE = r"copy_rows|cls_fix|cls_merge|compose_finish|copy_segments|ln_dparam|colsum_|embed_|im2col|gpt_embed|caption_targets|weighted_sum|gemm_small_m|__amd_rocclr"


def rows(path):
    out = []
    for ln in open(path):
        c = [x.strip() for x in ln.strip().strip("|").split("|")]
        if len(c) > 3 and c[0] and not set(c[0]) <= set("-") and c[0] not in ("kernel", "form"):
            out.append(c)
    return out


def main():
    shape_md, trace_md, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    print(f"# Floor table of the config-B step (present decomposition)\n")
    print(f"Inputs: `{shape_md}` (in-step GEMM launches, HIP events) and `{trace_md}` (kernel trace, {steps} steps, weight-gradient lane off).  "
          f"Floors: FLOPs / {MFMA / 1e15:.1f} PF for MFMA work, algorithmic bytes / {HBM / 1e12:.1f} TB/s for streaming work (the larger of the two where both apply).\n")
    print("## GEMM family, by shape\n")
    print("| form | M | N | K | epilogue | launches | measured ms/step | MFMA floor ms | HBM floor ms | floor ms | measured / floor |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    g_meas = g_floor = 0.0
    small = [0.0, 0.0, 0]
    for c in rows(shape_md):
        form, M, N, K, epi, n, us = c[0], int(c[1]), int(c[2]), int(c[3]), c[4], float(c[5]), float(c[6])
        extra = sum(1 for k in ("preact_out", "residual", "act_bwd_z") if k in epi)
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K + (1 + extra) * M * N)
        f_m, f_h = fl / MFMA * 1e6, by / HBM * 1e6
        fl_us = max(f_m, f_h)
        meas = n * us / 1e3
        g_meas += meas
        g_floor += n * fl_us / 1e3
        if meas < 0.3:
            small[0] += meas
            small[1] += n * fl_us / 1e3
            small[2] += int(n)
            continue
        print(f"| {form} | {M} | {N} | {K} | {epi} | {n:g} | {meas:.2f} | {n * f_m / 1e3:.2f} | {n * f_h / 1e3:.2f} | {n * fl_us / 1e3:.2f} | {meas / (n * fl_us / 1e3):.2f} |")
    print(f"| (the {small[2]} launches under 0.3 ms/step each) | | | | | {small[2]} | {small[0]:.2f} | | | {small[1]:.2f} | {small[0] / max(small[1], 1e-9):.2f} |")
    print(f"| **GEMM family** | | | | | | **{g_meas:.2f}** | | | **{g_floor:.2f}** | **{g_meas / g_floor:.2f}** |\n")
    print("## Everything else (kernel trace)\n")
    print("| family | launches/step | measured ms/step | floor us/launch | floor ms/step | measured / floor | the floor counts |")
    print("|---|---|---|---|---|---|---|")
    tr = rows(trace_md)
    used = set()
    o_meas = o_floor = 0.0
    for name, rx, fl_us, what in NON_GEMM:
        calls = tot = 0.0
        for i, c in enumerate(tr):
            if re.search(rx, c[0]):
                used.add(i)
                calls += float(c[1])
                tot += float(c[2])
        if not calls:
            continue
        meas = tot / steps
        n = calls / steps
        if fl_us is None:
            floor = meas
        else:
            # launches of smaller shapes (the abstractor's, the top decoder layer's window) are priced at the dominant shape's rate
            floor = min(meas, n * fl_us / 1e3)
        o_meas += meas
        o_floor += floor
        print(f"| {name} | {n:g} | {meas:.2f} | {'' if fl_us is None else f'{fl_us:.1f}'} | {floor:.2f} | {meas / floor if floor else float('inf'):.2f} | {what} |")
    glue = sum(float(c[2]) for i, c in enumerate(tr) if i not in used and re.search(GLUE, c[0])) / steps
    nglue = sum(float(c[1]) for i, c in enumerate(tr) if i not in used and re.search(GLUE, c[0])) / steps
    print(f"| glue (row copies, cls fix / merge, embedding sums, im2col, parameter-gradient finishes, ...) | {nglue:g} | {glue:.2f} | | {glue:.2f} | 1.00 | launch-bound: measured time taken as floor |")
    o_meas += glue
    o_floor += glue
    print(f"| **everything else** | | **{o_meas:.2f}** | | **{o_floor:.2f}** | **{o_meas / o_floor:.2f}** | |\n")
    print(f"**Step floor of the present decomposition: {g_floor:.1f} (GEMM) + {o_floor:.1f} (everything else) = {g_floor + o_floor:.1f} ms** against "
          f"{g_meas + o_meas:.1f} ms of kernel time per step in these inputs (the timed step is shorter than the serialised kernel sum by what the "
          f"weight-gradient lane overlaps).  north_star's 0.40 at step level is 61.0 ms.")


if __name__ == "__main__":
    main()
