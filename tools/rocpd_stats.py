"""Summarise a rocprofv3 rocpd sqlite (--kernel-trace) into a per-kernel table (like --stats CSV).
Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, (end - start) as dur from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(n, [0, 0, 1e30, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{short(n)}` | {a[0]} | {a[1]/1e6:.2f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {100*a[1]/tot:.1f} |")
    out = "\n".join(lines) + f"\n\ntotal kernel time {tot/1e6:.1f} ms over {len(rows)} dispatches\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
