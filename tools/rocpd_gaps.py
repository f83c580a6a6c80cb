#!/usr/bin/env python3
"""Device idle time between consecutive kernels of the timed prove, from a rocprofv3 --kernel-trace rocpd
database: total, histogram, and the idle attributed to the kernel that PRECEDES each gap (i.e. which
stage's host round trip the device was waiting for).

    python tools/rocpd_gaps.py run_results.db [--second-half] [--list-over MICROSECONDS] [--timeline FROM_MS TO_MS]

--list-over N: also lists every gap longer than N us in time order (offset into the prove, kernel before -> kernel after):
where the host does enough work between two launches for the device to notice.
"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    stream_c = next((c for c in ("stream_id", "queue_id", "stream") if c in cols), None)
    rows = db.execute(f"select {name_c}, start, end, {stream_c or '0'} from kernels order by start").fetchall()
    rows = [(n.replace("void ", "").replace("ovm::", "").split("(")[0], s, e, q) for n, s, e, q in rows if not n.startswith("at::") and "elementwise" not in n]
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
    streams = collections.defaultdict(lambda: [0, 0])
    for _, s0, e0, q in rows:
        streams[q][0] += 1
        streams[q][1] += e0 - s0
    if not stream_c:
        streams = {}
    rows = [(n, s0, e0) for n, s0, e0, _ in rows]
    # Kernels of different streams overlap (the early coset LDE runs on the library's side stream beside the GKR layers), so
    # "busy" is the UNION of the kernel intervals and a gap is a stretch in which NO kernel of any stream runs; it is
    # attributed to the kernel that ended last before it. (Round 3 summed per-kernel durations and consecutive start - end
    # differences: with overlap its headline and its histogram disagreed.)
    span = rows[-1][2] - rows[0][1]
    kernel_time = sum(e - s for _, s, e in rows)
    per = collections.defaultdict(lambda: [0, 0])
    hist = collections.Counter()
    list_over = float(sys.argv[sys.argv.index("--list-over") + 1]) * 1e3 if "--list-over" in sys.argv else None
    long_gaps = []
    busy, cur_end, last_name = 0, rows[0][1], rows[0][0]
    for n, s0, e0 in rows:
        if s0 > cur_end:
            g = s0 - cur_end
            per[last_name][0] += g
            per[last_name][1] += 1
            if list_over is not None and g > list_over:
                long_gaps.append((cur_end - rows[0][1], g, last_name, n))
            b = 0
            while (2 << b) * 1000 < g:
                b += 1
            hist[b] += g
            busy += e0 - s0
            cur_end, last_name = e0, n
        else:
            if e0 > cur_end:
                busy += e0 - cur_end
                cur_end, last_name = e0, n
    if "--timeline" in sys.argv:  # --timeline FROM_MS TO_MS: every kernel of the window (offset, gap before, duration, name)
        i = sys.argv.index("--timeline")
        lo, hi = float(sys.argv[i + 1]) * 1e6, float(sys.argv[i + 2]) * 1e6
        prev_end = None
        for n, s0, e0 in rows:
            off = s0 - rows[0][1]
            if lo <= off <= hi:
                gap = (s0 - prev_end) / 1e3 if prev_end is not None else 0.0
                print(f"  t {off / 1e6:9.4f} ms  gap {gap:8.1f} us  dur {(e0 - s0) / 1e3:8.1f} us  {n}")
            prev_end = max(prev_end, e0) if prev_end is not None else e0
    print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms "
          f"(sum of kernel durations {kernel_time / 1e6:.2f} ms: {max(kernel_time - busy, 0) / 1e6:.2f} ms ran concurrently)")
    if streams:
        print("per stream (kernels, time):")
        for sid, (c, t) in sorted(streams.items(), key=lambda kv: -kv[1][1]):
            print(f"  stream {sid}: {c:5d} kernels, {t / 1e6:7.2f} ms")
    print("idle by gap size:")
    for b in sorted(hist):
        print(f"  < {2 << b:5d} us : {hist[b] / 1e6:7.3f} ms")
    print("idle after kernel (top 15):")
    for n, (g, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:15]:
        print(f"  {n[:44]:44s} {g / 1e6:7.3f} ms over {c:4d} gaps ({g / max(c, 1) / 1e3:6.1f} us each)")


    if long_gaps:
        print(f"gaps over {list_over / 1e3:.0f} us in time order ({sum(g for _, g, _, _ in long_gaps) / 1e6:.3f} ms in {len(long_gaps)} gaps):")
        for at, g, a, b in long_gaps:
            print(f"  at {at / 1e6:7.3f} ms  {g / 1e3:7.1f} us   {a[:40]:40s} -> {b[:40]}")


if __name__ == "__main__":
    main()
