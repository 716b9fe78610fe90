"""Does the gradient all-reduce overlap the backward?  From a rocprofv3 --kernel-trace of `MPV_BENCH_FORCE_DIST=1 bench.py`
(RCCL path on one rank): every RCCL kernel with its queue / stream and the share of its duration during which a compute
kernel of ANOTHER queue was running.   Usage: rocpd_overlap.py DB [out.md]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:48]


def main():
    c = sqlite3.connect(sys.argv[1])
    ks = c.execute("select name, queue_id, stream_id, start, end from kernels order by start").fetchall()
    comm = [k for k in ks if re.search(r"nccl|rccl", k[0], re.I)]
    comp = [k for k in ks if not re.search(r"nccl|rccl", k[0], re.I)]
    lines = [f"{len(comm)} RCCL kernel launches on queues {sorted({k[1] for k in comm})} / streams {sorted({k[2] for k in comm})}; "
             f"{len(comp)} compute launches on queues {sorted({k[1] for k in comp})} / streams {sorted({k[2] for k in comp})}", "",
             "| RCCL kernel | queue | stream | duration us | overlapped by compute kernels of another queue | running beside it |", "|---|---|---|---|---|---|"]
    tot, cov = 0.0, 0.0
    j0 = 0
    for name, q, st, s, e in comm[-40:]:
        ov, names = 0, []
        for n2, q2, st2, s2, e2 in comp:
            if e2 <= s or s2 >= e or q2 == q:
                continue
            ov += min(e, e2) - max(s, s2)
            if len(names) < 3 and short(n2) not in names:
                names.append(short(n2))
        tot += e - s
        cov += min(ov, e - s)
        lines.append(f"| `{short(name)}` | {q} | {st} | {(e - s) / 1e3:.1f} | {100.0 * min(ov, e - s) / max(e - s, 1):.0f} % | {', '.join(names)} |")
    if tot > 0:
        lines.append(f"\nshare of RCCL kernel time (last {min(len(comm), 40)} launches) that ran under compute kernels of another queue: {100 * cov / tot:.0f} %")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
