"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database.  Usage: rocpd_counters.py DB [name-filter]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:60]


def main():
    c = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = {}
    for name, cname, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        k = short(name)
        if flt and flt not in k:
            continue
        a = agg.setdefault(k, {}).setdefault(cname, [0, 0.0])
        a[0] += 1
        a[1] += val
    for k, cs in sorted(agg.items()):
        print(k)
        for cn, (n, v) in sorted(cs.items()):
            print(f"    {cn:32s} launches {n:5d}  avg {v / n:16.1f}")


if __name__ == "__main__":
    main()
