"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes because
the TCC block has 4 counter slots).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B
requests of wide coalesced streams at 64 B -> doubled.  Usage: rocpd_pmc.py FETCH.db WRITE.db [out.md]"""
import os
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:70]


def load(db, counter):
    c = sqlite3.connect(db)
    agg = {}
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += val
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in f:
        n, fk = f[k]
        wk = w.get(k, [n, 0.0])[1]
        rows.append((2 * fk + wk, k, n, fk, wk))
    rows.sort(reverse=True)
    lines = ["| kernel | launches | FETCH_SIZE KB (raw) | WRITE_SIZE KB | HBM MB/launch (2*FETCH+WRITE) |", "|---|---|---|---|---|"]
    for tot, k, n, fk, wk in rows[:25]:
        lines.append(f"| `{k}` | {n} | {fk:.0f} | {wk:.0f} | {tot / n / 1024:.2f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out)
    if len(sys.argv) > 4:       # GEMM-family summary consumed by bench.py (roofline.traffic)
        import json
        gem = [(tot, n) for tot, k, n, fk, wk in rows if k.startswith("gemm_bf16_kernel") or k.startswith("gemm256_kernel")]
        nl = sum(n for _, n in gem)
        # per mpv_gemm_bf16 CALL (what bench.py's algorithmic bytes are per): a call can be several kernel launches (row bands);
        # the call count of the profiled process comes from its log (argv[5]), else fall back to kernel launches
        if len(sys.argv) > 5 and int(sys.argv[5]) > 0:
            nl = int(sys.argv[5])
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import gemm_source_digest
        json.dump({"hbm_mb_per_launch": round(sum(t for t, _ in gem) / nl / 1024, 1), "launches": nl, "per": "mpv_gemm_bf16 call" if len(sys.argv) > 5 and int(sys.argv[5]) > 0 else "kernel launch", "gemm_src_sha": gemm_source_digest(),
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1`, 2*FETCH+WRITE, profiles/" + os.path.basename(sys.argv[3])},
                  open(sys.argv[4], "w"))


if __name__ == "__main__":
    main()
