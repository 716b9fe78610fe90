"""Per-kernel matrix-pipe utilisation from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE
(+ optionally SQ_BUSY_CYCLES, SQ_WAVE_CYCLES).  On gfx950 SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the
cycles a SIMD's matrix pipe was busy (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32); GRBM_GUI_ACTIVE sums the active
cycles of the 8 XCDs.  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).   Usage: rocpd_mfma_util.py DB [out.md]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:64]


def main():
    c = sqlite3.connect(sys.argv[1])
    agg = {}
    for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        a = agg.setdefault(short(name), {}).setdefault(cname, [0, 0.0])
        a[0] += 1
        a[1] += val
    rows = []
    for k, cs in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
            continue
        n = cs["GRBM_GUI_ACTIVE"][0]
        busy, gui = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1], cs["GRBM_GUI_ACTIVE"][1]
        if busy <= 0:
            continue
        rows.append((gui, k, n, busy / n, gui / n, busy / (gui / 8.0 * 1024.0)))
    rows.sort(reverse=True)
    tot_busy = sum(r[3] * r[2] for r in rows if r[1].startswith("gemm"))
    tot_gui = sum(r[4] * r[2] for r in rows if r[1].startswith("gemm"))
    lines = ["| kernel | launches | MFMA busy cycles / launch (sum over 1024 SIMDs) | GRBM_GUI_ACTIVE / launch (sum over 8 XCDs) | MFMA pipe utilisation |",
             "|---|---|---|---|---|"]
    for gui, k, n, b, g, u in rows[:24]:
        lines.append(f"| `{k}` | {n} | {b:.3e} | {g:.3e} | {100 * u:.1f} % |")
    if tot_gui > 0:
        lines.append(f"\nGEMM family, all launches: MFMA pipe busy {100 * tot_busy / (tot_gui / 8.0 * 1024.0):.1f} % of the cycles the GPU was active in them")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
