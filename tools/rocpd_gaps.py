"""Where is the device idle inside a step?  From a rocprofv3 --kernel-trace rocpd sqlite of bench.py (weight-gradient lane off: one
stream, kernels do not overlap): the idle time between the end of a dispatch and the start of the next, summed per step and
broken down by the kernel that FOLLOWS the gap (the one whose launch / dependency resolution the device waited for).
Usage: python tools/rocpd_gaps.py <results.db> [out.md] [steps]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:70]


def main():
    db = sys.argv[1]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    # steady state: everything from the first adamw launch on (model construction and the first step's lazy work are before it)
    first = next((i for i, r in enumerate(rows) if "adamw" in r[0]), 0)
    rows = rows[first + 1:]
    nsteps = sum(1 for r in rows if "adamw" in r[0])
    busy = sum(e - s for _, s, e in rows)
    gaps, big = {}, 0.0
    idle = 0.0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g <= 0:
            continue
        if g > 200e3:            # host-side pauses between steps (bench bookkeeping), not launch gaps
            big += g
            continue
        idle += g
        a = gaps.setdefault(short(n1), [0, 0.0])
        a[0] += 1
        a[1] += g
    lines = [f"{len(rows)} dispatches over {nsteps} steady-state steps: busy {busy / 1e6 / max(nsteps, 1):.2f} ms/step, "
             f"idle between dispatches {idle / 1e6 / max(nsteps, 1):.2f} ms/step ({idle / max(len(rows), 1) / 1e3:.2f} us per dispatch on average); "
             f"pauses > 200 us (host bookkeeping between steps): {big / 1e6:.1f} ms in total", "",
             "| kernel after the gap | gaps | total idle ms | mean gap us |", "|---|---|---|---|"]
    for n, (k, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append(f"| `{n}` | {k} | {t / 1e6:.2f} | {t / k / 1e3:.2f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
