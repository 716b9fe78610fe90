"""Does the gradient all-reduce overlap the backward?  From a rocprofv3 --kernel-trace of an N >= 2 run (one rank's database):
every RCCL kernel with its queue / stream and the share of its duration during which a compute kernel of ANOTHER queue was
running (union of the compute intervals: two compute queues under one RCCL kernel are not counted twice).
Usage: rocpd_overlap.py DB [out.md]          (scripts/bench_scale.sh runs it on the N >= 2 traces)"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:48]


def covered(s, e, intervals):
    """Length of [s, e) covered by the union of `intervals` (list of (start, end))."""
    cut = sorted((max(s, a), min(e, b)) for a, b in intervals if b > s and a < e)
    total, cur_s, cur_e = 0, None, None
    for a, b in cut:
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        total += cur_e - cur_s
    return total


def overlap_report(kernels, last=40):
    """kernels: rows (name, queue_id, stream_id, start_ns, end_ns).  -> (markdown lines, share of RCCL kernel time under compute of
    another queue in [0, 1] or None without RCCL launches)."""
    ks = sorted(kernels, key=lambda k: k[3])
    comm = [k for k in ks if re.search(r"nccl|rccl", k[0], re.I)]
    comp = [k for k in ks if not re.search(r"nccl|rccl", k[0], re.I)]
    lines = [f"{len(comm)} RCCL kernel launches on queues {sorted({k[1] for k in comm})} / streams {sorted({k[2] for k in comm})}; "
             f"{len(comp)} compute launches on queues {sorted({k[1] for k in comp})} / streams {sorted({k[2] for k in comp})}", "",
             "| RCCL kernel | queue | stream | duration us | overlapped by compute kernels of another queue | running beside it |", "|---|---|---|---|---|---|"]
    tot, cov = 0.0, 0.0
    for name, q, st, s, e in comm[-last:]:
        beside = [(s2, e2) for n2, q2, st2, s2, e2 in comp if q2 != q and e2 > s and s2 < e]
        names = []
        for n2, q2, st2, s2, e2 in comp:
            if q2 != q and e2 > s and s2 < e and short(n2) not in names and len(names) < 3:
                names.append(short(n2))
        ov = covered(s, e, beside)
        tot += e - s
        cov += ov
        lines.append(f"| `{short(name)}` | {q} | {st} | {(e - s) / 1e3:.1f} | {100.0 * ov / max(e - s, 1):.0f} % | {', '.join(names)} |")
    share = cov / tot if tot > 0 else None
    if share is not None:
        lines.append(f"\nshare of RCCL kernel time (last {min(len(comm), last)} launches) that ran under compute kernels of another queue: {100 * share:.0f} %")
    return lines, share


def main():
    c = sqlite3.connect(sys.argv[1])
    ks = c.execute("select name, queue_id, stream_id, start, end from kernels order by start").fetchall()
    lines, _ = overlap_report(ks)
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
