#!/usr/bin/env python3
"""Device idle time between consecutive kernels of the timed prove, from a rocprofv3 --kernel-trace rocpd
database: total, histogram, and the idle attributed to the kernel that PRECEDES each gap (i.e. which
stage's host round trip the device was waiting for).

    python tools/rocpd_gaps.py run_results.db [--second-half]
"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_c}, start, end from kernels order by start").fetchall()
    rows = [(n.replace("void ", "").replace("ovm::", "").split("(")[0], s, e) for n, s, e in rows if not n.startswith("at::") and "elementwise" not in n]
    if "--second-half" in sys.argv:
        # bench.py --steps 1 --warmup 1 proves twice on the same input, so the trace ENDS with the same kernel sequence
        # twice: the timed prove is the longest suffix that repeats the kernels right before it (a cold first prove has
        # allocation gaps of milliseconds, so gap sizes are no guide)
        names = [r[0] for r in rows]
        best = 0
        for L in range(len(names) // 2, 50, -1):
            if names[-L:] == names[-2 * L:-L]:
                best = L
                break
        if best:
            rows = rows[-best:]
        else:
            # the first prove of a process also runs one-time initialisation kernels (fills of freshly allocated pages),
            # so the two sequences need not be equal: cut at the second launch of the first kernel that runs exactly once
            # per prove
            count = collections.Counter(names)
            first = next((i for i, n in enumerate(names) if count[n] == 2), None)
            if first is None:
                raise SystemExit("no kernel that runs once per prove: cannot find the timed prove")
            second = max(i for i, n in enumerate(names) if n == names[first])
            rows = rows[second:]
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    per = collections.defaultdict(lambda: [0, 0])
    hist = collections.Counter()
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = max(0, s1 - e0)
        per[n0][0] += g
        per[n0][1] += 1
        b = 0
        while (2 << b) * 1000 < g:
            b += 1
        hist[b] += g
    print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms")
    print("idle by gap size:")
    for b in sorted(hist):
        print(f"  < {2 << b:5d} us : {hist[b] / 1e6:7.3f} ms")
    print("idle after kernel (top 15):")
    for n, (g, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:15]:
        print(f"  {n[:44]:44s} {g / 1e6:7.3f} ms over {c:4d} gaps ({g / max(c, 1) / 1e3:6.1f} us each)")


if __name__ == "__main__":
    main()
