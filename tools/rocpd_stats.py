#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd sqlite database (--kernel-trace output).

    python tools/rocpd_stats.py run_results.db [--second-half] [--csv out.csv]

--second-half keeps, per kernel, only the later half of its dispatches (bench.py --steps 1
--warmup 1 runs the prove twice; the second one is the timed one).
"""
import collections
import sqlite3
import sys


def main():
    path = sys.argv[1]
    half = "--second-half" in sys.argv
    csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_c}, start, end from kernels order by start").fetchall()
    per = collections.defaultdict(list)
    for n, s, e in rows:
        n = n.replace("void ", "").replace("ovm::", "").split("(")[0]
        per[n].append(e - s)
    out = []
    for n, v in per.items():
        if half:
            v = v[len(v) // 2:]
        out.append((sum(v) / 1e6, len(v), sum(v) / len(v) / 1e3, max(v) / 1e3, n))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    lines = ["kernel,calls,total_ms,avg_us,max_us,percent"]
    for t, c, a, m, n in out:
        lines.append(f'"{n}",{c},{t:.3f},{a:.1f},{m:.1f},{100 * t / tot:.2f}')
    if csv:
        open(csv, "w").write("\n".join(lines) + "\n")
    print(f"total GPU ms: {tot:.2f}")
    for l in lines[:28]:
        print(l)


if __name__ == "__main__":
    main()
